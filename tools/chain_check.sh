#!/bin/bash
# One GPU call for a change to the factorisation chain (run through gpurun from the repo root): the Cholesky / linear-solve
# tests, the per-dispatch timeline of the chain (rocprofv3 kernel trace of tools/chol_timeline.py) and the default bench.
# usage: tools/chain_check.sh <tag> [full]      ("full": the whole GPU suite instead of the Cholesky subset)
set -uo pipefail
tag="$1"; out="gpurun_out/$tag"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
if [ "${2:-}" = "full" ]; then
  timeout -k 10 700 python3 -m pytest tests -m gpu -x -q > "$out/tests.log" 2>&1
else
  timeout -k 10 300 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "cholesky or linear or golden or large or indefinite" > "$out/tests.log" 2>&1
fi
rc=$?
tail -2 "$out/tests.log"
[ $rc -ne 0 ] && exit $rc
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d "$out/tl" -- python3 tools/chol_timeline.py run 7 1.0 > "$out/tl_run.log" 2>&1 &&
python3 tools/chol_timeline.py report "$out/tl" > "$out/chol_timeline.txt"
rm -rf "$out/tl"
grep -v "^#" "$out/chol_timeline.txt" | awk '{printf "%s:%s ", substr($2,1,12), $4} END {print ""}'
grep "^# span\|launches" "$out/chol_timeline.txt"
timeout -k 10 200 python3 bench.py --no-cpu-baseline --probe-L 0 > "$out/bench.log" 2> "$out/bench.err"
python3 - "$out/bench.log" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "ms_per_step", "newton_steps_per_solve")})
print(d["parity"]["z_rel_l2_vs_oracle"], d["roofline"]["kernel"], d["roofline"]["frac"], d["roofline"]["avg_launch_us"])
PY
