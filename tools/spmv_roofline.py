#!/usr/bin/env python3
"""Bandwidth of the barrier / SpMV kernels as a function of mesh size: HIP-event
timed back-to-back launches on the context stream (Amg::time_kernels), algorithmic bytes as in DESIGN.md
section 4a.  Shows where the kernels leave the launch-latency regime (L=7: everything cache-resident, ~10 us
per launch) and what fraction of the 8 TB/s HBM peak they reach on meshes that do not fit the caches.
usage: python3 tools/spmv_roofline.py [Lmin Lmax]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np          # noqa: E402
import mgb_amd as M         # noqa: E402

HBM_PEAK_GBS = 8000.0


INFINITY_CACHE_BYTES = 256 * 2 ** 20


def probe(L, p=1.0, reps=24, nrot=None):
    """nrot (default: as many distinct operand copies as it takes for the smallest rotating working set -- the barrier
    kernels' ~5 MB x L-dependent -- to exceed twice the 256 MiB Infinity Cache, capped at 8; 1 on cache-resident meshes
    means "same buffers every launch", which is how the solve itself runs at L=7)."""
    t0 = time.time()
    geo = M.fem2d_mpi(L)
    A = M.AMG(geo, p=p)
    x = geo.x.to_numpy()
    n = x.shape[0]
    c = np.tile(np.asarray(M.DEFAULT_F[2](x[0]), dtype=np.float64), (n, 1))
    z0 = np.column_stack([x[:, 0] ** 2 + x[:, 1] ** 2, np.full(n, 100.0)]).reshape(-1, order="F")   # DEFAULT_G[2]
    A.set_c(c)
    A.set_z(z0)
    if nrot is None:
        smallest = n * (2 * 4 + 3) * 8            # barrier_f0, the smallest launch
        nrot = 1 if L <= 7 else int(min(8, max(2, -(-2 * INFINITY_CACHE_BYTES // smallest))))
    kt = A.time_kernels(A.L - 1, reps, nrot)
    out = dict(L=L, n=n, N=A.level_size(A.L - 1)[0], setup_s=time.time() - t0, rotating_copies=nrot, kernels={})
    for k, v in kt.items():
        gbs = v["bytes"] / max(v["ms"], 1e-9) / 1e6
        out["kernels"][k] = dict(us=1e3 * v["ms"], MB=v["bytes"] / 1e6, GBs=gbs, frac_hbm=gbs / HBM_PEAK_GBS,
                                 rotating_working_set_MB=nrot * v["bytes"] / 1e6)
    # multigrid kernels of the same level: matrix-free H v, one Chebyshev step, transfers.
    # frac_hbm: the CSR-based algorithmic bytes SURVEY.md section 8(d) prescribes / time; frac_hbm_moved: what the kernel moves by
    # construction (values once for both halves, 1-byte local columns from cache-resident class tables) / time
    nrot_mg = 1 if L <= 7 else int(min(8, max(2, -(-2 * INFINITY_CACHE_BYTES // max(1, n * 200)))))
    out["mg_rotating_copies"] = nrot_mg
    out["mg_kernels"] = {}
    for k, v in A.time_mg_kernels(A.L - 1, reps, nrot_mg).items():
        if v["ms"] <= 0:
            out["mg_kernels"][k] = dict(MB=v["bytes"] / 1e6)
            continue
        out["mg_kernels"][k] = dict(us=1e3 * v["ms"], MB_moved=v["bytes"] / 1e6, MB_algorithmic=v["algorithmic_bytes"] / 1e6,
                                    frac_hbm=v["algorithmic_bytes"] / v["ms"] / 1e6 / HBM_PEAK_GBS,
                                    frac_hbm_moved=v["bytes"] / v["ms"] / 1e6 / HBM_PEAK_GBS)
    return out


if __name__ == "__main__":
    lo = int(sys.argv[1]) if len(sys.argv) > 1 else 7
    hi = int(sys.argv[2]) if len(sys.argv) > 2 else 9
    for L in range(lo, hi + 1):
        print(json.dumps(probe(L)), flush=True)
