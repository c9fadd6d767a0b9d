#!/usr/bin/env python3
"""Where a Newton step of the timed solve spends its time, from a rocprofv3 kernel trace of bench.py.

  run:      rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl -- python3 bench.py --steps 1 --no-cpu-baseline --probe-L 0
  analyse:  python3 tools/solve_timeline.py gpurun_out/tl > profiles/rN_solve_timeline.txt
Splits the dispatch stream of the LAST solve into Newton steps (a step starts at the first factorisation kernel after a
non-factorisation kernel) and reports, as medians over the steps: kernel time and idle gaps of the factorisation chain, of the
launches between two chains, and the largest gaps (host round trips)."""
import csv
import glob
import os
import statistics
import sys


def short(name):
    for k in ("front_leaf", "front_single", "front_start", "front_step", "front_panel2", "front_update2", "backward_rect", "backward_kernel", "trial_f0", "barrier_f0",
              "barrier_f1", "barrier_f2", "spmv_kernel", "elop_assemble", "gather_sum", "elop_apply", "dof_gather", "csr_apply", "dot_kernel", "sum_kernel", "final_sum", "waxpby", "copyBuffer", "fillBuffer",
              "front_top"):
        if k in name:
            return k
    return name[:24]


def main(d):
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])))
    rows.sort()
    chol = {"front_leaf", "front_single", "front_start", "front_step", "front_panel2", "front_update2", "backward_rect", "backward_kernel", "front_top"}
    # Newton steps: maximal runs [chain][other kernels]
    steps, cur, in_chain = [], [], False
    for s, e, k in rows:
        c = k in chol
        if c and not in_chain and cur:
            steps.append(cur)
            cur = []
        in_chain = c
        cur.append((s, e, k))
    if cur:
        steps.append(cur)
    steps = [st for st in steps if sum(1 for r in st if r[2] in chol) >= 8]
    steps = steps[len(steps) // 2:]      # the timed solve (the warm-up solve comes first)
    tot, chain_k, chain_gap, other_k, other_gap, big = [], [], [], [], [], []
    prev_end = None
    for st in steps:
        t0 = st[0][0]
        t1 = st[-1][1]
        ck = sum(e - s for s, e, k in st if k in chol)
        ok = sum(e - s for s, e, k in st if k not in chol)
        cg = og = 0
        gaps = []
        for (s0, e0, k0), (s1, e1, k1) in zip(st, st[1:]):
            g = max(0, s1 - e0)
            if k0 in chol and k1 in chol:
                cg += g
            else:
                og += g
            gaps.append((g, k0, k1))
        lead = (st[0][0] - prev_end) if prev_end else 0
        prev_end = t1
        tot.append((t1 - t0 + lead) / 1e3)
        chain_k.append(ck / 1e3); chain_gap.append(cg / 1e3); other_k.append(ok / 1e3); other_gap.append((og + lead) / 1e3)
        gaps.sort(reverse=True)
        big.append(gaps[:3])
    med = statistics.median
    print("# %d Newton steps of the last solve; medians per step (us)" % len(steps))
    print("step wall            %8.1f" % med(tot))
    print("chain kernels        %8.1f" % med(chain_k))
    print("chain gaps           %8.1f" % med(chain_gap))
    print("other kernels        %8.1f" % med(other_k))
    print("other gaps (host)    %8.1f" % med(other_gap))
    # host gaps by position: median and mean over the steps of the idle time before each kind of non-chain launch, and of the
    # lead before the chain (the step graph's launch)
    from collections import defaultdict
    by = defaultdict(list)
    pe = None
    for st in steps:
        for s, e, k in st:
            if pe is not None and (k not in chol or k == st[0][2] and s == st[0][0]):
                by[("chain start" if k in chol else k)].append(max(0, s - pe) / 1e3)
            pe = e
    print("# idle time before a launch, by kind of the launch: count median mean (us)")
    for k, v in sorted(by.items(), key=lambda kv: -sum(kv[1])):
        print("gap before %-16s %6d %7.2f %7.2f" % (k, len(v), med(v), sum(v) / len(v)))
    # typical sequence of one step
    st = steps[len(steps) // 2]
    print("# one step, dispatch by dispatch: kernel dur_us gap_before_us")
    pe = None
    for s, e, k in st:
        print("%-16s %7.2f %7.2f" % (k, (e - s) / 1e3, 0.0 if pe is None else (s - pe) / 1e3))
        pe = e


if __name__ == "__main__":
    main(sys.argv[1])
