#!/bin/bash
# Build libmgb_hip.so for gfx950 (MI355X).  hipcc cross-compiles without a GPU.
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
OUT="$HERE/../lib"
mkdir -p "$OUT" "$HERE/_obj"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
CXXFLAGS="-O3 -std=c++17 -fPIC -mavx2 -mfma -Wall -Wno-unused-result"
for f in geometry mfchol; do
  if [ "$HERE/$f.cpp" -nt "$HERE/_obj/$f.o" ] || [ "$HERE/sparse.hpp" -nt "$HERE/_obj/$f.o" ] || [ "$HERE/$f.hpp" -nt "$HERE/_obj/$f.o" ]; then
    g++ $CXXFLAGS -c "$HERE/$f.cpp" -o "$HERE/_obj/$f.o" &
  fi
done
"$HIPCC" --offload-arch=gfx950 $CXXFLAGS -c "$HERE/kernels.hip" -o "$HERE/_obj/kernels.o" &
"$HIPCC" --offload-arch=gfx950 $CXXFLAGS -c "$HERE/gpuchol.hip" -o "$HERE/_obj/gpuchol.o" &
"$HIPCC" --offload-arch=gfx950 $CXXFLAGS -x hip -c "$HERE/amg.cpp" -o "$HERE/_obj/amg.o" &
"$HIPCC" --offload-arch=gfx950 $CXXFLAGS -x hip -c "$HERE/capi.cpp" -o "$HERE/_obj/capi.o" &
wait
"$HIPCC" --offload-arch=gfx950 -shared -fPIC -o "$OUT/libmgb_hip.so" "$HERE"/_obj/{geometry,mfchol,kernels,gpuchol,amg,capi}.o -lpthread
echo "built $OUT/libmgb_hip.so"
