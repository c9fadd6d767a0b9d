"""Diagnostic: device estimate of lambda_max(Dinv H) against scipy's, and CG iteration counts, per level."""
import sys, os
import numpy as np, scipy.sparse as sp, scipy.sparse.linalg as spla
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import mgb_amd as M
from test_gpu_parity import _problem

kind, L, p = sys.argv[1], int(sys.argv[2]), float(sys.argv[3])
A, Mo, B, z0, c, go = _problem(M, kind, L, p)
rng = np.random.default_rng(4)
for l in range(1, L):
    N = A.level_size(l)[0]
    s = np.zeros(N)
    H, _ = A.f2(l, s, 10.0)
    d = H.diagonal()
    lam = float(spla.eigsh(sp.diags(d ** -0.5) @ H @ sp.diags(d ** -0.5), k=1, which="LA", return_eigenvectors=False)[0])
    b = rng.standard_normal(N)
    for its in (6, 12, 24):
        A.set_pcg(power_its=its)
        _, est = A.smooth(l, s, b, degree=2, sweeps=1, lmax=0.0)
        print("level %d N=%d lambda_max %.4f  device estimate (%d power steps) %.4f  ratio %.3f" % (l, N, lam, its, est, est / lam))
    g = A.f1(l, s, 10.0)
    A.set_pcg(power_its=6, rtol=1e-10, maxit=200)
    x, it, rr, ok = A.pcg_solve_linear(l, s, g)
    print("   pcg: %d iterations, resid %.2e, converged %s, |Hx-g|/|g| %.2e" % (it, rr, ok, np.linalg.norm(H @ x - g) / np.linalg.norm(g)))
