#!/bin/bash
# One environment knob of the factorisation chain compared over several values in one GPU call: per value the per-dispatch
# timeline of the chain (rocprofv3 kernel trace of tools/chol_timeline.py at fem2d L=7).
# usage (through gpurun, from the repo root): tools/env_sweep.sh <tag> <VARIABLE> <value> [<value> ...]
set -uo pipefail
out="gpurun_out/$1"; var="$2"; shift 2; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for v in "$@"; do
  env "$var=$v" true
  export "$var=$v"
  timeout -k 10 120 rocprofv3 --kernel-trace --output-format csv -d "$out/tl_$v" -- python3 tools/chol_timeline.py run 7 1.0 > "$out/run_$v.log" 2>&1 || { echo "$var=$v failed"; tail -3 "$out/run_$v.log"; exit 1; }
  python3 tools/chol_timeline.py report "$out/tl_$v" > "$out/timeline_${var}_$v.txt"
  rm -rf "$out/tl_$v"
  echo "$var=$v: $(grep residual "$out/run_$v.log")"
  grep -v "^#" "$out/timeline_${var}_$v.txt" | awk '{printf "%s:%s ", substr($2,7,4), $4} END {print ""}'
  grep "^# span" "$out/timeline_${var}_$v.txt"
done
