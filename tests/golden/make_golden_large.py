"""Generates the BASELINE-size golden fixtures tests/golden/large_*.npz from the CPU oracle
(oracle/mgb_oracle.py) -- NOT from the reference: Julia is absent from the build image and the
reference stores no solve outputs, so these vectors pin the oracle/HIP pair at the
headline sizes ("parity unpinned" at solve level).  The oracle needs minutes per case here, which is
why the GPU box compares against these files instead of running it.

    python tests/golden/make_golden_large.py fem2d 7 1.0
    python tests/golden/make_golden_large.py fem2d 7 1.5
    python tests/golden/make_golden_large.py fem3d 4 1.0 mmd     (fill-reducing LU ordering, see use_fill_reducing_ordering)
    python tests/golden/make_golden_large.py parabolic 6 1.0     (h = 0.1, t1 = 1: snapshots 1, 5, 10)

Only z (and the scalar histories) are stored: fem2d L=7 z is 57 344 x 2 doubles."""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
import mgb_oracle as O  # noqa: E402

PARABOLIC_KEEP = (1, 5, 10)


def use_fill_reducing_ordering():
    """Optional 4th argument `mmd`: the oracle's A \\ b (scipy SuperLU) with the column ordering MMD_AT_PLUS_A instead of the
    default COLAMD.  Same solver, same arithmetic per entry; for the symmetric 3-D Newton matrices (fem3d L=4: 28 k unknowns,
    ~700 couplings per row) it cuts the fill, and with it the hours the default ordering needs, by an order of magnitude."""
    import scipy.sparse as sp
    import scipy.sparse.linalg as spla

    def solve(H, g):
        H = sp.csc_matrix(H)
        if H.shape[0] == 0:
            return np.zeros(0)
        return spla.splu(H, permc_spec="MMD_AT_PLUS_A").solve(g)

    O.solve = solve


def main():
    kind, L, p = sys.argv[1], int(sys.argv[2]), float(sys.argv[3])
    if len(sys.argv) > 4 and sys.argv[4] == "mmd":
        use_fill_reducing_ordering()
    tag = "%s_L%d_p%s" % (kind, L, str(p).replace(".", "_"))
    t0 = time.time()
    if kind == "parabolic":
        g = O.fem2d(L)
        sol = O.parabolic_solve(g, h=0.1, t1=1.0, p=p)
        out = dict(ts=sol.ts, keep=np.array(PARABOLIC_KEEP))
        for k in PARABOLIC_KEEP:
            out["u_%d" % k] = sol.u[k]
        info = "snapshots %d" % len(sol.u)
    else:
        g = getattr(O, kind)(L)
        sol = O.amgb(g, p=p)
        out = dict(z=sol.z, its=sol.SOL_main["its"], ts=sol.SOL_main["ts"], c_dot_Dz=sol.SOL_main["c_dot_Dz"])
        info = "steps %d |z| %.17g" % (int(sol.SOL_main["its"].sum()), np.linalg.norm(sol.z))
    out["oracle_seconds"] = time.time() - t0
    out["oracle_lu_ordering"] = np.array("MMD_AT_PLUS_A" if (len(sys.argv) > 4 and sys.argv[4] == "mmd") else "COLAMD")
    np.savez_compressed(os.path.join(HERE, "large_%s.npz" % tag), **out)
    print(tag, info, "oracle %.0f s" % (time.time() - t0), flush=True)


if __name__ == "__main__":
    main()
