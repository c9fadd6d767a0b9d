"""Adds `z_centre` to a headline-size golden file: the oracle's end point polished to the EXACT centre of the last barrier
parameter t_final, and reports how far any other end point (e.g. one saved from the HIP path) is from it.

Why: Newton on the finest level stops on stagnation (`stopping_exact`: the objective, ~1e9 at t = 1e8, no longer decreases in
double precision).  That rule resolves the centre only to the last Newton step it did not take; whether that step is taken
depends on the rounding of the objective sum, so two faithful implementations end up to one (tiny) Newton step apart -- for
fem3d L=4 the oracle stops 2.0e-10 short, the HIP path 1.9e-13.  The polished point removes the stop rule from the comparison:
the gradient of t c.Dz + w.F(Dz) is evaluated in 80-bit extended precision (x87 long double: D z, the cone terms and D'
all accumulated in long double), Newton systems are solved with the oracle's own Hessian and LU in double, and the iteration
is repeated until the step is below 1e-14 |z|.  Everything here is oracle code (numpy/scipy on the CPU); nothing from the HIP
path enters `z_centre`.

    python tests/golden/polish_centre.py tests/golden/large_fem3d_L4_p1_0.npz [other_end_point.npz ...]
"""
import os
import re
import sys

import numpy as np
import scipy.sparse as sp

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
import mgb_oracle as O  # noqa: E402

LD = np.longdouble


def spmv_ld(A, v):
    A = sp.coo_matrix(A)
    out = np.zeros(A.shape[0], dtype=LD)
    np.add.at(out, A.row, A.data.astype(LD) * v[A.col])
    return out


class Problem:
    def __init__(self, kind, L, p, t):
        g = getattr(O, kind)(L)
        self.M = O.amg(g)
        self.dim = g.discretization["dim"]
        self.p, self.t = p, t
        self.c = O.map_rows(O.DEFAULT_F[self.dim], self.M.x)
        self.B = O.Barrier(O.convex_Euclidian_power(idx=list(range(1, self.dim + 2)), p=p))
        self.n = self.M.x.shape[0]

    def newton_step(self, z):
        """R d with H d = gradient(z): gradient in long double, H and the LU in double (the oracle's own solve)."""
        M, dim = self.M, self.dim
        R, D, w = M.R[-1], M.D, M.w
        zv = z.reshape(-1, order="F")
        Dz = np.stack([spmv_ld(Dk, zv.astype(LD)) for Dk in D], axis=1)
        q, s = Dz[:, 1:dim + 1], Dz[:, dim + 1]
        a, mu = LD(2) / LD(self.p), LD(O.barrier_mu(self.p))
        phi = np.power(s, a) - np.sum(q * q, axis=1)
        assert float(phi.min()) > 0 and float(s.min()) > 0
        G = np.zeros_like(Dz)
        G[:, 1:dim + 1] = 2 * q / phi[:, None]
        G[:, dim + 1] = -a * np.power(s, a - 1) / phi - mu / s
        y = G + LD(self.t) * self.c.astype(LD)
        ret = np.zeros(D[0].shape[1], dtype=LD)
        for k in range(len(D)):
            ret += spmv_ld(D[k].T, w.astype(LD) * y[:, k])
        grad = np.asarray(spmv_ld(R.T, ret), dtype=np.float64)
        H = sp.csc_matrix(self.B.f2(np.zeros(R.shape[1]), M.x, w, self.t * self.c, R, D, zv))
        d = R @ O.solve(H, grad)
        return d.reshape(z.shape, order="F"), float(phi.min())


def rel(a, b):
    return float(np.linalg.norm(a) / np.linalg.norm(b))


def main():
    path = sys.argv[1]
    m = re.match(r"large_(fem\dd)_L(\d+)_p(\d+)_(\d+)\.npz", os.path.basename(path))
    kind, L, p = m.group(1), int(m.group(2)), float(m.group(3) + "." + m.group(4))
    gold = dict(np.load(path))
    P = Problem(kind, L, p, float(gold["ts"][-1]))
    z = gold["z"].copy()
    for it in range(6):
        d, phimin = P.newton_step(z)
        print("polish %d: min phi %.3e  |step|/|z| %.3e  (u %.3e  s %.3e)" % (
            it, phimin, rel(d, z), rel(d[:, 0], z[:, 0]), rel(d[:, 1], z[:, 1])), flush=True)
        z = z - d
        if rel(d, z) < 1e-14:
            break
    gold["z_centre"] = z
    gold["oracle_end_point_to_centre"] = rel(gold["z"] - z, z)
    np.savez_compressed(path, **gold)
    print("%s: oracle end point is %.3e from the centre" % (os.path.basename(path), gold["oracle_end_point_to_centre"]))
    for other in sys.argv[2:]:
        zo = np.load(other)["z"]
        print("%s: %.3e from the centre (u %.3e  s %.3e); %.3e from the oracle end point" % (
            os.path.basename(other), rel(zo - z, z), rel(zo[:, 0] - z[:, 0], z[:, 0]), rel(zo[:, 1] - z[:, 1], z[:, 1]),
            rel(zo - gold["z"], z)))


if __name__ == "__main__":
    main()
