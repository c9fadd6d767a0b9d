#!/usr/bin/env python3
"""solver="pcg" beside the direct solver (CG iterations per Newton step and us per step at L=7/8/9).
For each mesh: the direct solve; the pcg solve with the give-up rule (CG while it converges, direct afterwards); and the pcg solve
that keeps trying (giveup=0) on the smaller meshes.  usage: python3 tools/pcg_vs_direct.py [L ...]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np          # noqa: E402
import mgb_amd as M         # noqa: E402

Ls = [int(a) for a in sys.argv[1:]] or [7, 8, 9]
print("%-18s %-26s %7s %8s %9s %7s %9s %10s %10s %9s" % ("mesh", "solver", "newton", "solve s", "us/newton", "CG sys", "CG iters", "its/system", "us/CG it", "gave up"))
for L in Ls:
    for p in (1.5, 1.0):
        zref = None
        for name, kw in (("direct", dict()),
                         ("pcg (give up after 3)", dict(solver="pcg", pcg=dict(rtol=1e-8, maxit=100))),
                         ("pcg (keeps trying)", dict(solver="pcg", pcg=dict(rtol=1e-8, maxit=100, giveup=0)))):
            if name.endswith("trying)") and (L > 7 or p == 1.0):
                continue
            t0 = time.time()
            s = M.fem2d_mpi_solve(L=L, p=p, **kw)
            z = M.mpi_to_native(s).z
            S = s.SOL_main
            nn = int(S["its"].sum())
            pc = S["pcg"]
            err = "" if zref is None else " |dz| %.1e" % (np.linalg.norm(z - zref) / np.linalg.norm(zref))
            zref = z if zref is None else zref
            print("%-18s %-26s %7d %8.3f %9.1f %7d %9d %10.1f %10.1f %9s%s" % (
                "fem2d L=%d p=%g" % (L, p), name, nn, S["t_elapsed"], 1e6 * S["t_elapsed"] / nn, pc["solves"], pc["iterations"],
                pc["iterations"] / max(pc["solves"], 1), 1e6 * pc["seconds"] / max(pc["iterations"], 1),
                "-" if pc["gave_up_at"] < 0 else "at %d" % pc["gave_up_at"], err), flush=True)
