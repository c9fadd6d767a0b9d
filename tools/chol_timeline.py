#!/usr/bin/env python3
"""Dispatch timeline of ONE device factorisation + solve (GpuChol) from a rocprofv3 kernel trace.

  run:      rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl -- python3 tools/chol_timeline.py run 7 1.0
  analyse:  python3 tools/chol_timeline.py report gpurun_out/tl > gpurun_out/timeline.txt
The report lists every dispatch of a factor+solve chain (kernel, workgroups, duration, idle gap since the previous
dispatch ended; medians over the chains of the run) and the per-kernel totals."""
import csv
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(L, p, reps=24, kind="fem2d"):
    import numpy as np
    import mgb_amd as M
    geo = getattr(M, kind + "_mpi")(L)
    A = M.AMG(geo, p=p)
    dim = {"fem1d": 1, "fem2d": 2, "fem3d": 3}[kind]
    x = geo.x.to_numpy()
    A.set_c(np.vstack([M.DEFAULT_F[dim](xi) for xi in x]))
    A.set_z(np.vstack([M.DEFAULT_G[dim](xi) for xi in x]).reshape(-1, order="F"))
    l = A.L - 1
    N = A.level_size(l)[0]
    H, lower = A.f2(l, np.zeros(N), 0.1)
    g = A.f1(l, np.zeros(N), 0.1)
    for _ in range(reps):
        xs = A.solve_linear(l, lower, g)
    r = H @ xs - g
    print("N=%d residual %.3e" % (N, np.linalg.norm(r) / np.linalg.norm(g)))


def report(d):
    files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    rows = []
    for f in files:
        with open(f) as fh:
            for r in csv.DictReader(fh):
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"],
                             int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0), int(r.get("Workgroup_Size_X", 1) or 1)))
    rows.sort()
    names = ("front_leaf_kernel", "front_single_dense_kernel", "front_single_kernel", "front_start_kernel", "front_step_kernel", "front_step2_kernel", "front_panel2_kernel", "front_update2_kernel", "backward_rect_kernel", "backward_kernel")

    def short(s):
        for nm in names:
            if nm in s:
                return nm
        return s.split("(")[0][-28:]

    # every factor+solve chain: from a factorisation launch that does not follow another one to the last backward_kernel
    # before the next chain; chains with the launch count of the last one are kept (the first one also builds the graph)
    ks = [short(r[2]) for r in rows]
    fact = ("front_leaf_kernel", "front_start_kernel", "front_single_kernel", "front_single_dense_kernel", "front_step_kernel", "front_step2_kernel", "front_panel2_kernel", "front_update2_kernel")
    starts = [i for i, k in enumerate(ks) if k in fact[:2] and (i == 0 or ks[i - 1] not in fact)]
    chains = []
    for a, b in zip(starts, starts[1:] + [len(rows)]):
        last = max((i for i in range(a, b) if ks[i] == "backward_kernel"), default=None)
        if last is not None:
            chains.append(rows[a:last + 1])
    chains = [c for c in chains if len(c) == len(chains[-1])][1:] or chains[-1:]
    med = lambda v: sorted(v)[len(v) // 2]
    tot, gaps = {}, 0.0
    print("# %d chains; median per dispatch slot" % len(chains))
    print("# idx kernel workgroups dur_us gap_us")
    for j in range(len(chains[0])):
        s, e, name, grid, wg = chains[-1][j]
        dur = med([(c[j][1] - c[j][0]) / 1e3 for c in chains])
        gap = 0.0 if j == 0 else med([(c[j][0] - c[j - 1][1]) / 1e3 for c in chains])
        gaps += max(gap, 0.0)
        k = short(name)
        t = tot.setdefault(k, [0, 0.0])
        t[0] += 1
        t[1] += dur
        print("%4d %-28s %6d %8.2f %7.2f" % (j, k, grid // max(wg, 1), dur, gap))
    print("# span %.1f us, kernel time %.1f us, idle gaps %.1f us over %d dispatches" %
          (med([(c[-1][1] - c[0][0]) / 1e3 for c in chains]), sum(v[1] for v in tot.values()), gaps, len(chains[0])))
    for k, (c, t) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
        print("# %-28s %4d launches %9.1f us  avg %7.2f" % (k, c, t, t / c))


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(int(sys.argv[2]), float(sys.argv[3]), kind=sys.argv[4] if len(sys.argv) > 4 else "fem2d")
    else:
        report(sys.argv[2])
