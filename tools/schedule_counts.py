#!/usr/bin/env python3
"""Newton steps, time and end point of the two level schedules of amgb_step on the GPU: "fine" (default: Newton on the finest
subspace only) and "all" (the literal coarse -> fine loop R_1 .. R_L of SURVEY.md section 3.1; docs/src/guide.md:158 counts
`sum(SOL_main.its)` over that loop).  usage: python3 tools/schedule_counts.py [Lmin Lmax]  ->  profiles/rN_schedule_newton_counts.txt"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np          # noqa: E402
import mgb_amd as M         # noqa: E402

lo = int(sys.argv[1]) if len(sys.argv) > 1 else 5
hi = int(sys.argv[2]) if len(sys.argv) > 2 else 7
print("# fem2d, tol = sqrt(eps); per-level Newton steps of schedule 'all' listed coarse -> fine")
for L in range(lo, hi + 1):
    for p in (1.5, 1.0):
        out = {}
        for sched in ("fine", "all"):
            M.fem2d_mpi_solve(L=min(L, 3), p=p, schedule=sched)          # code objects
            t0 = time.time()
            try:
                sol = M.fem2d_mpi_solve(L=L, p=p, schedule=sched)
            except M.MGBError as exc:
                print("L=%d p=%g %-4s FAILED: %s" % (L, p, sched, exc), flush=True)
                continue
            out[sched] = (M.mpi_to_native(sol).z, sol.SOL_main["its"], time.time() - t0)
            its = out[sched][1]
            print("L=%d p=%g %-4s newton %5d  per level %s  wall %.2f s (incl. setup)" % (
                L, p, sched, int(its.sum()), its.sum(axis=1).tolist(), out[sched][2]), flush=True)
        if len(out) == 2:
            zf, za = out["fine"][0], out["all"][0]
            print("L=%d p=%g  |z_all - z_fine| / |z_fine| = %.3e" % (L, p, np.linalg.norm(za - zf) / np.linalg.norm(zf)), flush=True)
