import sys, time
sys.path.insert(0, "/root/repo/oracle")
import numpy as np, scipy.sparse as sp, scipy.sparse.linalg as spla
import mgb_oracle as O

L, p, tstop = int(sys.argv[1]), float(sys.argv[2]), float(sys.argv[3])
g = O.fem2d(L)
M = O.amg(g)
x, w = M.x, M.w
n = x.shape[0]
z = O.map_rows(lambda xi: O.DEFAULT_G[2](xi), x).reshape(-1, order="F")
c = O.map_rows(lambda xi: O.DEFAULT_F[2](xi), x)
Q = O.convex_Euclidian_power([1, 2, 3], p)
B = O.Barrier(Q)
tol = np.sqrt(np.finfo(float).eps)
# follow the path until t >= tstop
state = {}
def es(Dz0):
    return state.get("t", 0) >= tstop
orig_step = O.amgb_step
def step(B_, M_, z_, Dz0_, c_, *a, **k):
    r = orig_step(B_, M_, z_, Dz0_, c_, *a, **k)
    state["t"] = c_[0, 3] / c[0, 3]
    state["z"] = r["z"]
    return r
O.amgb_step = step
SOL = O.amgb_core(B, M, z, c, tol, early_stop=es)
zt, t = state["z"], state["t"]
print("reached t=%g" % t)
R = M.R[-1]
Dsp = M.D
Dz = B.apply_D(Dsp, zt)
Y = Q.F2(x, Dz)           # n x K x K
H = O.hessian_recipe(Dsp, w, Y, R).tocsr()
grad = B.f1(np.zeros(R.shape[1]), x, w, t * c, R, Dsp, zt)
N = H.shape[0]
# unknown split: u = dirichlet block first, s = full block
Nu = g.subspaces["dirichlet"][-1].shape[1]
iu, isl = np.arange(Nu), np.arange(Nu, N)
Auu, Aus, Ass = H[iu][:, iu], H[iu][:, isl], H[isl][:, isl]
offd = abs(Ass - sp.diags(Ass.diagonal())).max()
print("N=%d Nu=%d  A_ss offdiag max %.1e" % (N, Nu, offd))
dss = Ass.diagonal()
S = (Auu - Aus @ sp.diags(1 / dss) @ Aus.T).tocsc()
# node-wise condensed: eliminate s (row 3) from Y
Yc = Y[:, 1:3, 1:3] - Y[:, 1:3, 3:4] * Y[:, 3:4, 1:3] / Y[:, 3:4, 3:4]
Ru = g.subspaces["dirichlet"][-1]
dx, dy = g.operators["dx"], g.operators["dy"]
Bx, By = dx @ Ru, dy @ Ru
St = (Bx.T @ sp.diags(w * Yc[:, 0, 0]) @ Bx + By.T @ sp.diags(w * Yc[:, 1, 1]) @ By
      + Bx.T @ sp.diags(w * Yc[:, 0, 1]) @ By + By.T @ sp.diags(w * Yc[:, 0, 1]) @ Bx).tocsc()
def pcg(A, b, Minv, rtol=1e-8, maxit=500):
    xk = np.zeros_like(b); r = b.copy(); zk = Minv(r); pk = zk.copy(); rz = r @ zk; rz0 = rz
    for k in range(maxit):
        Ap = A @ pk; al = rz / (pk @ Ap); xk += al * pk; r -= al * Ap
        zk = Minv(r); rzn = r @ zk
        if rzn <= rtol ** 2 * rz0: return xk, k + 1
        pk = zk + (rzn / rz) * pk; rz = rzn
    return xk, maxit
# 1. exact condensed solve as preconditioner for S
lu = spla.splu(St)
bu = grad[iu] - Aus @ (grad[isl] / dss)
_, it = pcg(S, bu, lu.solve)
print("PCG on exact Schur S with exact node-condensed S~ solve: %d iterations" % it)
ev = spla.eigsh(S, k=1, M=St, which="LA", return_eigenvectors=False)[0]
ev2 = spla.eigsh(S, k=1, M=St, sigma=0, which="LM", return_eigenvectors=False)[0]
print("   spectrum of S~^-1 S in [%.3g, %.3g]" % (ev2, ev))
# 2. MG V-cycle (Galerkin, Chebyshev-Jacobi deg 2) on S~, geometric hierarchy of the dirichlet subspaces
def prolong(Rf, Rc):
    Rf = Rf.tocsr(); rep = -np.ones(Rf.shape[1], dtype=int)
    for r in range(Rf.shape[0]):
        s, e = Rf.indptr[r], Rf.indptr[r + 1]
        if e - s == 1 and abs(Rf.data[s] - 1) < 1e-12 and rep[Rf.indices[s]] < 0: rep[Rf.indices[s]] = r
    return Rc.tocsr()[rep]
Ps = [prolong(g.subspaces["dirichlet"][l + 1], g.subspaces["dirichlet"][l]) for l in range(L - 1)]
def make_vcycle(Afine, Ps, deg=2, lo=0.12, hi=1.2, cmin=1):
    As = [None] * (len(Ps) + 1); As[-1] = Afine.tocsr()
    for l in range(len(Ps) - 1, -1, -1): As[l] = (Ps[l].T @ As[l + 1] @ Ps[l]).tocsr()
    lam = [None] * len(As)
    for l in range(cmin + 1, len(As)):
        d = As[l].diagonal()
        lam[l] = spla.eigsh(sp.diags(d ** -.5) @ As[l] @ sp.diags(d ** -.5), k=1, which="LA", return_eigenvectors=False)[0]
    cl = spla.splu(As[cmin].tocsc())
    def cheb(A, b, x0, lm):
        dinv = 1 / A.diagonal(); h, lo_ = hi * lm, lo * lm; th, de = (h + lo_) / 2, (h - lo_) / 2; sg = th / de
        xx = x0.copy(); r = b - A @ xx; rho = 1 / sg; d = dinv * r / th
        for k in range(1, deg):
            xx += d; r -= A @ d; rn = 1 / (2 * sg - rho); d = rn * rho * d + 2 * rn / de * dinv * r; rho = rn
        return xx + d
    def V(l, b):
        if l == cmin: return cl.solve(b)
        xx = cheb(As[l], b, np.zeros_like(b), lam[l])
        r = b - As[l] @ xx
        xx += Ps[l - 1] @ V(l - 1, Ps[l - 1].T @ r)
        return cheb(As[l], b, xx, lam[l])
    return lambda b: V(len(As) - 1, b)
for deg in (2, 3):
    Vc = make_vcycle(St, Ps, deg=deg)
    _, it = pcg(St, bu, Vc)
    print("PCG on S~ with V-cycle (deg %d): %d iterations" % (deg, it))
    _, it = pcg(S, bu, Vc)
    print("PCG on exact S with V-cycle on S~ (deg %d): %d iterations" % (deg, it))
# 3. reference: plain V-cycle on the full system (what the GPU does now)
Pfull = [sp.block_diag([Ps[l], prolong(g.subspaces["full"][l + 1], g.subspaces["full"][l])]).tocsr() for l in range(L - 1)]
Vf = make_vcycle(H, Pfull, deg=2)
_, it = pcg(H, grad, Vf)
print("PCG on full H with plain V-cycle: %d iterations" % it)
print("nnz S %d, nnz A_uu %d" % (S.nnz, Auu.nnz))
for deg in (2, 4):
    Vs = make_vcycle(S, Ps, deg=deg)
    _, it = pcg(S, bu, Vs)
    print("PCG on exact S with Galerkin V-cycle on S itself (deg %d): %d iterations" % (deg, it))
# two-level with exact coarse solve one level down, to separate smoother quality from coarse-space quality
Vs2 = make_vcycle(S, Ps, deg=2, cmin=len(Ps) - 1)
_, it = pcg(S, bu, Vs2)
print("PCG on exact S, TWO-level (exact solve on level L-2), deg 2: %d iterations" % it)
# vertex-patch additive Schwarz smoother on S (blocks = u dofs sharing an s dof = star of a node), as a one-level preconditioner + coarse
Sd = S.tocsr()
pat = (abs(Aus) > 0).tocsc()
blocks = [np.unique(pat.indices[pat.indptr[k]:pat.indptr[k + 1]]) for k in range(pat.shape[1])]
blocks = [b for b in blocks if len(b)]
inv = [np.linalg.inv(Sd[b][:, b].toarray()) for b in blocks]
cnt = np.zeros(Nu)
for b in blocks: cnt[b] += 1
def patch(r):
    z = np.zeros_like(r)
    for b, Bi in zip(blocks, inv): z[b] += Bi @ r[b]
    return z / cnt.max()
Pu = Ps[-1]
Sc = spla.splu((Pu.T @ S @ Pu).tocsc())
def two_level_patch(r):
    z = patch(r)
    r2 = r - S @ z
    z = z + Pu @ Sc.solve(Pu.T @ r2)
    r3 = r - S @ z
    return z + patch(r3)
_, it = pcg(S, bu, two_level_patch)
print("PCG on exact S, node-star patch smoother + exact coarse (two-level): %d iterations (max overlap %d, mean block %d)" % (it, cnt.max(), np.mean([len(b) for b in blocks])))
