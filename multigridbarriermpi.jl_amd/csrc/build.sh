#!/bin/bash
# Build libmgb_hip.so for gfx950 (MI355X).  hipcc cross-compiles without a GPU.
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
OUT="$HERE/../lib"
mkdir -p "$OUT" "$HERE/_obj"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
CXXFLAGS="-O3 -std=c++17 -fPIC -mavx2 -mfma -Wall -Wno-unused-result"
# kernel arguments preloaded into SGPRs at wave launch (gfx940+): the chain kernels start with descriptor -> data, not
# argument pointer -> descriptor -> data
PRELOAD="-mllvm -amdgpu-kernarg-preload-count=16"
HDRS=("$HERE"/*.hpp "$HERE/../../include/mgb_hip.h")
pids=()
# compile <object> <command...>: rebuild when the source or any header is newer; a stale object never survives a failed
# compile (it is deleted first), and every background job's exit status is checked
compile() {
  local obj="$1" src="$2"; shift 2
  local stale=0
  [ -f "$obj" ] || stale=1
  if [ $stale -eq 0 ]; then
    for f in "$src" "${HDRS[@]}"; do [ "$f" -nt "$obj" ] && stale=1; done
  fi
  if [ $stale -eq 1 ]; then
    rm -f "$obj"
    "$@" -c "$src" -o "$obj" &
    pids+=($!)
  fi
}
compile "$HERE/_obj/geometry.o" "$HERE/geometry.cpp" g++ $CXXFLAGS
compile "$HERE/_obj/mfchol.o" "$HERE/mfchol.cpp" g++ $CXXFLAGS
compile "$HERE/_obj/kernels.o" "$HERE/kernels.hip" "$HIPCC" --offload-arch=gfx950 $CXXFLAGS
compile "$HERE/_obj/kernels_f32.o" "$HERE/kernels_f32.hip" "$HIPCC" --offload-arch=gfx950 $CXXFLAGS
compile "$HERE/_obj/mg.o" "$HERE/mg.hip" "$HIPCC" --offload-arch=gfx950 $CXXFLAGS
compile "$HERE/_obj/amg_mg.o" "$HERE/amg_mg.cpp" "$HIPCC" --offload-arch=gfx950 $CXXFLAGS -x hip
compile "$HERE/_obj/gpuchol.o" "$HERE/gpuchol.hip" "$HIPCC" --offload-arch=gfx950 $CXXFLAGS $PRELOAD
compile "$HERE/_obj/amg.o" "$HERE/amg.cpp" "$HIPCC" --offload-arch=gfx950 $CXXFLAGS -x hip
compile "$HERE/_obj/comm.o" "$HERE/comm.cpp" "$HIPCC" --offload-arch=gfx950 $CXXFLAGS -x hip
compile "$HERE/_obj/capi.o" "$HERE/capi.cpp" "$HIPCC" --offload-arch=gfx950 $CXXFLAGS -x hip
for pid in "${pids[@]}"; do
  wait "$pid" || { echo "build.sh: a compile job failed" >&2; exit 1; }
done
rm -f "$OUT/libmgb_hip.so"
"$HIPCC" --offload-arch=gfx950 -shared -fPIC -o "$OUT/libmgb_hip.so" "$HERE"/_obj/{geometry,mfchol,kernels,kernels_f32,mg,gpuchol,amg,amg_mg,comm,capi}.o -lpthread -ldl
echo "built $OUT/libmgb_hip.so"
