#!/bin/bash
# Profiles of one round on the GPU box (run through gpurun from the repo root): rocprofv3 kernel stats of the default
# bench, the two PMC passes (FETCH_SIZE / WRITE_SIZE, separate runs, kernel trace only) of the same command, and the same
# two passes of the rotating-buffer bandwidth probe at L=9.  Writes summaries to gpurun_out/$1/.
set -uo pipefail
out="gpurun_out/$1"
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
CMD="python3 bench.py --no-cpu-baseline --probe-L 0"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -- $CMD > "$out/bench_stats_run.log" 2>&1
f=$(find "$out/stats" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$out/kernel_stats_L7_p1.csv"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$out/pmc_$c" -- $CMD > "$out/bench_pmc_$c.log" 2>&1
  f=$(find "$out/pmc_$c" -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 tools/pmc_summary.py "$f" > "$out/pmc_${c}_per_kernel.csv"
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$out/probe_$c" -- python3 tools/spmv_roofline.py 9 9 > "$out/probe_pmc_$c.log" 2>&1
  f=$(find "$out/probe_$c" -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 tools/pmc_summary.py "$f" > "$out/probe_L9_pmc_${c}_per_kernel.csv"
done
python3 tools/pmc_to_json.py "$out/pmc_FETCH_SIZE_per_kernel.csv" "$out/pmc_WRITE_SIZE_per_kernel.csv" "fem2d L=7 p=1" > "$out/pmc_traffic.json"
rm -rf "$out/stats" "$out"/pmc_FETCH_SIZE "$out"/pmc_WRITE_SIZE "$out"/probe_FETCH_SIZE "$out"/probe_WRITE_SIZE
ls -la "$out"
