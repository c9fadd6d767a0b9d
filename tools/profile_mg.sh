#!/bin/bash
# rocprofv3 passes of the multigrid kernel probe (tools/mg_kernel_probe.py) on the GPU box: kernel stats + the two PMC passes
# (FETCH_SIZE / WRITE_SIZE, separate runs, kernel trace only) at L=7 (same buffers) and L=9 (3 rotating copies).  Summaries go to
# gpurun_out/$1/; every profiler run is bounded by a timeout.
set -uo pipefail
out="gpurun_out/$1"
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for L in 7 9; do
  CMD="python3 tools/mg_kernel_probe.py $L 24"
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/st$L" -- $CMD > "$out/mg_probe_L${L}.log" 2>&1
  f=$(find "$out/st$L" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$out/mg_kernel_stats_L${L}.csv"
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$out/pmc$L$c" -- $CMD > "$out/mg_pmc_L${L}_$c.log" 2>&1
    f=$(find "$out/pmc$L$c" -name "*counter_collection.csv" | head -1)
    [ -n "$f" ] && python3 tools/pmc_summary.py "$f" > "$out/mg_pmc_L${L}_${c}_per_kernel.csv"
  done
  rm -rf "$out/st$L" "$out"/pmc$L*
done
ls -la "$out"
