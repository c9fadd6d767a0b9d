#!/usr/bin/env python3
"""Back-to-back launches of the multigrid kernels (matrix-free H v, Chebyshev step, assembled H v, transfers) on fem2d level L
with rotating operand copies -- the command the rocprofv3 kernel-trace / PMC passes of profiles/r3_mg_* run.
usage: python3 tools/mg_kernel_probe.py L [reps] [nrot]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np          # noqa: E402
import mgb_amd as M         # noqa: E402

L = int(sys.argv[1])
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 24
nrot = int(sys.argv[3]) if len(sys.argv) > 3 else (1 if L <= 7 else 3)
geo = M.fem2d_mpi(L)
A = M.AMG(geo, p=1.0)
x = geo.x.to_numpy()
n = x.shape[0]
A.set_c(np.tile(np.asarray(M.DEFAULT_F[2](x[0]), dtype=np.float64), (n, 1)))
A.set_z(np.column_stack([x[:, 0] ** 2 + x[:, 1] ** 2, np.full(n, 100.0)]).reshape(-1, order="F"))
out = dict(L=L, n=n, N=A.level_size(A.L - 1)[0], reps=reps, rotating_copies=nrot, kernels={})
for k, v in A.time_mg_kernels(A.L - 1, reps, nrot).items():
    if v["ms"] > 0:
        out["kernels"][k] = dict(us=1e3 * v["ms"], MB_moved=v["bytes"] / 1e6, MB_algorithmic=v["algorithmic_bytes"] / 1e6,
                                 frac_hbm=v["algorithmic_bytes"] / v["ms"] / 1e6 / 8000.0,
                                 frac_hbm_moved=v["bytes"] / v["ms"] / 1e6 / 8000.0)
print(json.dumps(out))
