"""Generates tests/golden/*.npz from the CPU oracle (oracle/mgb_oracle.py) -- NOT from the reference:
Julia is absent from the build image and the reference stores no solve outputs, so
these vectors pin the oracle/HIP pair against regressions, not against the reference ("parity
unpinned" at solve level).  Run from the repo root:  python tests/golden/make_golden.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
import mgb_oracle as O  # noqa: E402

CASES = [("fem1d", 3, 1.0), ("fem1d", 4, 2.0), ("fem2d", 2, 1.5), ("fem2d", 3, 1.0), ("fem2d", 3, 2.0),
         ("fem3d", 2, 1.0), ("fem3d", 2, 2.0)]          # fem3d: k = 3 (reference default)


def main():
    for kind, L, p in CASES:
        g = getattr(O, kind)(L)                       # fem3d: default k = 3
        dim = g.discretization["dim"]
        sol = O.amgb(g, p=p)
        M = O.amg(g)
        B = O.Barrier(O.convex_Euclidian_power(list(range(1, dim + 2)), p))
        x = M.x
        z0 = O.map_rows(lambda xi: O.DEFAULT_G[dim](xi), x).reshape(-1, order="F")
        c = O.map_rows(lambda xi: O.DEFAULT_F[dim](xi), x)
        out = dict(z=sol.z, its=sol.SOL_main["its"], ts=sol.SOL_main["ts"], c_dot_Dz=sol.SOL_main["c_dot_Dz"])
        rng = np.random.default_rng(7)
        for l in range(L):
            R = M.R[l]
            s = 1e-3 * rng.normal(size=R.shape[1])
            out["s_%d" % l] = s
            out["f0_%d" % l] = B.f0(s, x, M.w, 2.5 * c, R, M.D, z0)
            out["f1_%d" % l] = B.f1(s, x, M.w, 2.5 * c, R, M.D, z0)
            H = B.f2(s, x, M.w, 2.5 * c, R, M.D, z0)
            out["f2diag_%d" % l] = H.diagonal()
            out["f2fro_%d" % l] = np.sqrt((H.multiply(H)).sum())
        name = "%s_L%d_p%s.npz" % (kind, L, str(p).replace(".", "_"))
        np.savez_compressed(os.path.join(HERE, name), **out)
        print(name, "steps", int(sol.SOL_main["its"].sum()), "|z|", np.linalg.norm(sol.z))


if __name__ == "__main__":
    main()
