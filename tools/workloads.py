#!/usr/bin/env python3
"""Wall times of the BASELINE.json configurations (and a few neighbours) on one GPU, one JSON line each.
usage: python3 tools/workloads.py [quick | l9]      (l9: only the two fem2d L=9 solves, 917 504 rows)"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np          # noqa: E402
import mgb_amd as M         # noqa: E402


def solve(kind, L, p, **kw):
    t0 = time.time()
    sol = getattr(M, kind + "_mpi_solve")(L=L, p=p, **kw)
    M.backend_hip(0).synchronize()
    wall = time.time() - t0
    main = sol.SOL_main
    n = sol.z.shape[0]
    its = int(np.sum(main["its"]))
    print(json.dumps(dict(workload="%s L=%d p=%g" % (kind, L, p), n=n, newton_steps=its, solve_s=main["t_elapsed"],
                          wall_incl_setup_s=wall, dof_per_s_per_step=n * its / main["t_elapsed"],
                          linear_solve_s=main["time_factor"])), flush=True)


def parabolic(L, p, h):
    g = M.fem2d_mpi(L)
    t0 = time.time()
    sol = M.parabolic_solve(g, h=h, t1=1.0, p=p)
    M.backend_hip(0).synchronize()
    print(json.dumps(dict(workload="parabolic fem2d L=%d p=%g h=%g" % (L, p, h), steps=len(sol.ts) - 1,
                          wall_s=time.time() - t0)), flush=True)


if __name__ == "__main__":
    quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
    solve("fem2d", 3, 1.0)                     # configs[0] (plumbing case) -- also warms the code objects
    if len(sys.argv) > 1 and sys.argv[1] == "l9":
        solve("fem2d", 9, 1.5)
        solve("fem2d", 9, 1.0)
        sys.exit(0)
    solve("fem2d", 5, 1.5)                     # configs[1]
    solve("fem2d", 7, 1.0)                     # configs[2] workload on one GPU (bench.py default)
    solve("fem2d", 7, 1.5)
    solve("fem2d", 7, 2.0)
    solve("fem3d", 4, 1.0)                     # configs[3]
    parabolic(6, 2.0, 0.1)                     # configs[4]
    if not quick:
        parabolic(6, 1.0, 0.1)
        solve("fem2d", 8, 1.0)
        solve("fem1d", 12, 1.0)
