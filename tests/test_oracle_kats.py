"""Pins the CPU oracle against every known-answer test the reference holds for this path
(SURVEY.md §8c items 1-6).  Citations are into /root/reference (not read at run time)."""
import numpy as np
import scipy.sparse as sp
import pytest

import mgb_oracle as O


# ---- 1. map_rows exact values (test/test_helpers.jl:129-167, test/test_map_rows.jl:27-101)
def test_map_rows_helpers_kats():
    r = O.map_rows(lambda row: np.sum(row), np.array([[1., 2], [3, 4], [5, 6]]))
    assert r.tolist() == [3.0, 7.0, 11.0]
    r = O.map_rows(lambda row: np.array([np.sum(row), np.prod(row)]), np.array([[1., 2], [3, 4]]))
    assert r.tolist() == [[3.0, 2.0], [7.0, 12.0]]
    r = O.map_rows(lambda rx, ry: np.sum(rx) + ry[0], np.array([[1., 2], [3, 4]]), np.array([10., 20]))
    assert r.tolist() == [13.0, 27.0]


def test_map_rows_file_kats():
    v = np.arange(1.0, 9.0)
    assert np.array_equal(O.map_rows(lambda x: x[0] ** 2, v), v ** 2)
    w = v[::-1].copy()
    assert np.array_equal(O.map_rows(lambda x, y: x[0] * y[0], v, w), v * w)
    assert np.array_equal(O.map_rows(lambda x: np.array([x[0], x[0] ** 2, x[0] ** 3]), v), np.stack([v, v ** 2, v ** 3], 1))
    m = np.arange(1.0, 17.0).reshape(8, 2, order="F")          # reshape(1:16, 8, 2) is column-major in Julia
    assert np.array_equal(O.map_rows(lambda x: np.sum(x) ** 2, m), m.sum(axis=1) ** 2)


# ---- 2. small hooks (test/test_helpers.jl:53-121, test/test_diag.jl:28-46)
def test_small_hooks():
    assert O.amgb_all_isfinite([1.0, 2.0, 3.0]) is True
    assert O.amgb_all_isfinite([1.0, np.inf, 3.0]) is False
    assert O.amgb_zeros(5, 5).shape == (5, 5) and O.amgb_zeros(5, 5).nnz == 0
    assert O.amgb_diag(np.arange(1.0, 4.0)).shape == (3, 3)
    D = O.amgb_diag(np.arange(1.0, 11.0))
    assert (D != sp.diags(np.arange(1.0, 11.0))).nnz == 0
    B = O.amgb_blockdiag(sp.identity(2), sp.identity(3))
    assert B.shape == (5, 5) and (B != sp.identity(5)).nnz == 0


# ---- 3. sparse algebra (test/test_basic_ops.jl:27-97)
def test_basic_sparse_ops():
    A = sp.csr_matrix(np.array([[1., 0], [2, 3], [0, 4]]))
    B = sp.csr_matrix(np.array([[1., 2, 3], [4, 5, 6]]))
    assert np.array_equal((A @ B).toarray(), np.array([[1., 2, 3], [14, 19, 24], [16, 20, 24]]))
    AtA = (A.T @ A).toarray()
    assert np.array_equal(AtA, np.array([[5., 6], [6, 25]]))
    x = O.solve(sp.csr_matrix(AtA + 0.01 * np.eye(2)), np.ones(2))
    assert np.allclose((AtA + 0.01 * np.eye(2)) @ x, np.ones(2), atol=1e-10)


# ---- 4. Hessian recipe identity (test/test_matrix_addition.jl:39-95, test/test_d0_construction.jl:92-185)
@pytest.mark.parametrize("L", [2, 3])
def test_hessian_recipe_identity_fem1d(L):
    g = O.fem1d(L)
    n = g.x.shape[0]
    dx, ident = g.operators["dx"], g.operators["id"]
    Z = sp.csr_matrix((n, n))
    D0_dx, D0_id = sp.hstack([dx, Z], format="csr"), sp.hstack([Z, ident], format="csr")
    assert D0_dx.shape == (n, 2 * n)                        # hcat(dx, Z) is 8x16 at L=2 (test_partition_debug.jl:34)
    y11, y12, y22 = 0.5, 0.1, 0.3
    w = g.w
    y = np.zeros((n, 2, 2))
    y[:, 0, 0], y[:, 1, 0], y[:, 0, 1], y[:, 1, 1] = y11, y12, y12, y22
    H = O.hessian_recipe([D0_dx, D0_id], w, y)
    # explicit dense restatement
    Wd = lambda v: np.diag(w * v)
    Hd = (D0_dx.T @ Wd(y11) @ D0_dx + D0_id.T @ Wd(y22) @ D0_id + D0_dx.T @ Wd(y12) @ D0_id + D0_id.T @ Wd(y12) @ D0_dx)
    assert np.abs(H.toarray() - Hd).max() < 1e-12
    R = g.subspaces["dirichlet"][-1]
    Rb = sp.block_diag([R, R], format="csr")                # test_d0_construction.jl:82
    RHR = O.hessian_recipe([D0_dx, D0_id], w, y, Rb)
    assert np.abs(RHR.toarray() - Rb.T.toarray() @ Hd @ Rb.toarray()).max() < 1e-12
    ev = np.linalg.eigvalsh(RHR.toarray())
    assert ev.min() > 0                                      # SPD (test_hessian.jl:96-100)


# ---- 5. structure (test/test_nonsquare.jl:28, test_partition_debug.jl:34, docs/src/guide.md:246-253)
def test_structure_fem1d():
    g = O.fem1d(3)
    assert g.x.shape[0] == 16 and g.subspaces["dirichlet"][-1].shape == (16, 7)
    assert O.fem1d(2).x.shape[0] == 8
    for l in range(3):
        assert abs(g.coarsen[l] @ g.refine[l] - sp.identity(g.refine[l].shape[1])).max() < 1e-14


@pytest.mark.parametrize("L,n", [(1, 14), (2, 56), (3, 224), (4, 896), (5, 3584)])
def test_structure_fem2d_sizes(L, n):
    g = O.fem2d(L)
    assert g.x.shape == (n, 2)
    assert set(g.operators) == {"id", "dx", "dy"}            # tools/profile_ops.jl:35-37
    assert set(g.subspaces) == {"dirichlet", "full"}
    assert abs(g.w.sum() - 4.0) < 1e-12                       # area of [-1,1]^2
    assert g.operators["dx"].nnz <= 7 * n


def test_fem2d_counts_match_survey_table():
    g = O.fem2d(5)                                            # L=5 full 1601, dirichlet 1473
    assert g.subspaces["full"][-1].shape[1] == 1601
    assert g.subspaces["dirichlet"][-1].shape[1] == 1473


# ---- PDE-level known answers for the parts no reference vector pins
def test_fem2d_operators_exact_on_p2():
    g = O.fem2d(3)
    x, y = g.x[:, 0], g.x[:, 1]
    assert np.abs(g.operators["dx"] @ (x * x) - 2 * x).max() < 1e-12
    assert np.abs(g.operators["dy"] @ (x * y) - x).max() < 1e-12
    assert abs(np.dot(g.w, x ** 2 * y ** 2) - 4.0 / 9.0) < 1e-3   # cubic-exact rule, O(h^4) on quartics
    for l in range(3):
        assert abs(g.coarsen[l] @ g.refine[l] - sp.identity(g.refine[l].shape[1])).max() < 1e-13


def test_subspaces_are_nested_and_continuous():
    g = O.fem2d(3)
    for key in ("full", "dirichlet"):
        for l in range(2):
            Pc, Pf = g.subspaces[key][l], g.subspaces[key][l + 1]
            # every coarse basis function is reproduced by the finer space
            for j in range(0, Pc.shape[1], 3):
                col = Pc[:, j].toarray().ravel()
                co = sp.linalg.lsqr(Pf, col, atol=1e-14, btol=1e-14)[0]
                assert np.abs(Pf @ co - col).max() < 1e-10


@pytest.mark.parametrize("p", [1.0, 1.5, 2.0, 3.0])
def test_barrier_derivatives_match_finite_differences(p):
    rng = np.random.default_rng(0)
    Q = O.convex_Euclidian_power(idx=[1, 2, 3], p=p)
    Y = rng.normal(size=(50, 4))
    Y[:, 3] = (np.sum(Y[:, 1:3] ** 2, axis=1)) ** (p / 2) + rng.uniform(0.1, 2.0, 50)
    F1, F2 = Q.F1(None, Y), Q.F2(None, Y)
    h = 1e-6
    for k in range(4):
        E = np.zeros((1, 4)); E[0, k] = h
        fd = (Q.F(None, Y + E) - Q.F(None, Y - E)) / (2 * h)
        assert np.abs(fd - F1[:, k]).max() < 1e-5 * (1 + np.abs(F1[:, k]).max())
        fd2 = (Q.F1(None, Y + E) - Q.F1(None, Y - E)) / (2 * h)
        assert np.abs(fd2 - F2[:, :, k]).max() < 1e-4 * (1 + np.abs(F2).max())
    assert np.linalg.eigvalsh(F2[:, 1:, 1:]).min() > 0       # convex on the cone interior
    assert np.all(F2[:, 0, :] == 0)                           # the barrier ignores Dz[:,0] (u itself)


def test_newton_pieces_consistent():
    """f1 is the gradient and f2 the Hessian of f0 on a coarse subspace (restated algebra, §8a a4-a6)."""
    g = O.fem2d(2)
    M = O.amg(g)
    x = M.x
    z0 = O.map_rows(lambda xi: O.DEFAULT_G[2](xi), x).reshape(-1, order="F")
    c = O.map_rows(lambda xi: O.DEFAULT_F[2](xi), x)
    B = O.Barrier(O.convex_Euclidian_power([1, 2, 3], 1.5))
    R = M.R[0]
    rng = np.random.default_rng(1)
    s = 0.01 * rng.normal(size=R.shape[1])
    g1 = B.f1(s, x, M.w, c, R, M.D, z0)
    H = B.f2(s, x, M.w, c, R, M.D, z0).toarray()
    h = 1e-6
    for j in range(R.shape[1]):
        e = np.zeros_like(s); e[j] = h
        fd = (B.f0(s + e, x, M.w, c, R, M.D, z0) - B.f0(s - e, x, M.w, c, R, M.D, z0)) / (2 * h)
        assert abs(fd - g1[j]) < 1e-5 * (1 + abs(g1[j]))
        fdg = (B.f1(s + e, x, M.w, c, R, M.D, z0) - B.f1(s - e, x, M.w, c, R, M.D, z0)) / (2 * h)
        assert np.abs(fdg - H[:, j]).max() < 1e-4 * (1 + np.abs(H).max())


# ---- solve-level properties ("parity unpinned": no reference numbers exist; these are invariants)
@pytest.mark.parametrize("p", [1.0, 2.0])
def test_fem1d_solve_properties(p):
    sol = O.fem1d_solve(L=3, p=p)
    z = sol.z
    assert z.shape == (16, 2)
    g = sol.geometry
    # boundary data kept, constraint s >= |u'|^p strictly satisfied, objective decreasing along t
    assert abs(z[0, 0] + 1) < 1e-14 and abs(z[-1, 0] - 1) < 1e-14
    du = g.operators["dx"] @ z[:, 0]
    assert np.all(z[:, 1] > np.abs(du) ** p)
    cd = sol.SOL_main["c_dot_Dz"]
    assert np.all(np.diff(cd) < 1e-9)
    assert sol.SOL_main["ts"][-1] > 1 / np.sqrt(np.finfo(float).eps)
    # at the optimum the slack is active somewhere up to O(1/t) (s is continuous P1, |u'| is
    # piecewise constant, so it cannot be tight everywhere)
    assert (z[:, 1] - np.abs(du) ** p).min() < 1e-4


def test_fem2d_solution_independent_of_linesearch_path():
    a = O.fem2d_solve(L=2, p=1.0).z
    old = O.BETA
    try:
        O.BETA = 0.25
        b = O.fem2d_solve(L=2, p=1.0).z
    finally:
        O.BETA = old
    assert np.linalg.norm(a - b) / np.linalg.norm(a) < 1e-10


@pytest.mark.parametrize("L,k", [(1, 3), (2, 1), (2, 3)])
def test_structure_fem3d(L, k):
    """fem3d: Q_k hexahedra, (k+1)^3 nodes per element (src:682-684), operator keys incl. :dz (src:736)."""
    g = O.fem3d(L, k)
    n = 8 ** (L - 1) * (k + 1) ** 3
    assert g.x.shape == (n, 3) and abs(g.w.sum() - 8.0) < 1e-12 and g.w.min() > 0
    assert set(g.operators) == {"id", "dx", "dy", "dz"}
    x, y, z = g.x.T
    deg = min(k, 2)
    f = x ** deg * y + y * z
    assert np.abs(g.operators["dx"] @ f - deg * x ** (deg - 1) * y).max() < 1e-11
    assert np.abs(g.operators["dy"] @ f - (x ** deg + z)).max() < 1e-11
    assert np.abs(g.operators["dz"] @ f - y).max() < 1e-11
    for l in range(L):
        assert abs(g.coarsen[l] @ g.refine[l] - sp.identity(g.refine[l].shape[1])).max() < 1e-12
    npts = k * 2 ** (L - 1) + 1
    assert g.subspaces["full"][-1].shape == (n, npts ** 3)
    assert g.subspaces["dirichlet"][-1].shape == (n, (npts - 2) ** 3)


def test_fem3d_solve_properties():
    """src:735-745 defaults: D = [u id; u dx; u dy; u dz; s id], f = (.5,0,0,0,1), g = (|x|^2, 100)."""
    sol = O.fem3d_solve(L=2, k=2, p=1.5)
    z, g = sol.z, sol.geometry
    assert z.shape == (216, 2)
    grad2 = sum((g.operators[k] @ z[:, 0]) ** 2 for k in ("dx", "dy", "dz"))
    assert np.all(z[:, 1] > grad2 ** 0.75)
    bnd = np.any(np.abs(np.abs(g.x) - 1.0) < 1e-14, axis=1)
    assert np.abs(z[bnd, 0] - np.sum(g.x[bnd] ** 2, axis=1)).max() < 1e-13
    assert np.all(np.diff(sol.SOL_main["c_dot_Dz"]) < 1e-9)


def test_feasibility_phase_oracle():
    """An infeasible start is repaired (SOL_feasibility is a record, not None: src:428-455) and the main phase
    reaches the same z as from the default feasible start."""
    bad = O.amgb(O.fem2d(2), p=1.5, g=lambda x: np.array([x[0] ** 2 + x[1] ** 2, 0.5]))
    ok = O.amgb(O.fem2d(2), p=1.5)
    assert bad.SOL_feasibility is not None and ok.SOL_feasibility is None
    assert np.linalg.norm(bad.z - ok.z) / np.linalg.norm(ok.z) < 1e-9


def test_oracle_general_feasibility_phase():
    """The slack feasibility phase of the oracle (amgb_phase1_slack; [UPSTREAM-UNVERIFIED] in its details, SOL_feasibility
    src:428-455): from a start outside both the obstacle and the cone it stops early with a negative slack, the main phase
    then runs to t = 1/tol, and an impossible obstacle is reported."""
    import pytest
    g = O.fem1d(4)
    f = lambda x: np.array([0.5, 0.0, 1.0])
    gg = lambda x: np.array([1.0 - 1.5 * (1.0 - float(x[0] ** 2)), 0.05])
    sol = O.amgb(g, p=1.5, f=f, g=gg, extra=[O.LinearBarrier([0], [1.0], 0.2)])
    F = sol.SOL_feasibility
    assert F is not None and 2 <= len(F["ts"]) < 6 and F["sigma0"] > 1.0          # stopped early: t never got near 1/tol
    assert sol.SOL_main["ts"][-1] > 1e7 and sol.z[:, 0].min() > -0.2
    q = g.operators["dx"] @ sol.z[:, 0]
    assert np.all(sol.z[:, 1] > np.abs(q) ** 1.5)
    with pytest.raises(RuntimeError):
        O.amgb(g, p=1.5, f=f, g=gg, extra=[O.LinearBarrier([0], [1.0], -5.0)])


def test_both_continuation_stop_rules_visit_the_nominal_sequence():
    """the end of the t-continuation is selectable (oracle STOP_RULE / mgb_amg_set_stop_rule): "upstream" = the
    literal `while t <= 1/tol: t <- kappa t`, "fixed" = stop at the first t0 kappa^k beyond 1/tol.
    SOL_main.ts is an observable of the reference (docs/src/api.md:97-101); when kappa is never reduced both rules must give
    exactly ts = t0 kappa^k, k = 0 .. with the last one the first beyond 1/tol, and the same z."""
    sols = {r: O.amgb(O.fem1d(3), p=2.0, stop_rule=r) for r in ("fixed", "upstream")}
    tol = np.sqrt(np.finfo(np.float64).eps)
    want = [0.1]
    while want[-1] <= 1 / tol:
        want.append(want[-1] * 10.0)
    for r, s in sols.items():
        assert np.array_equal(s.SOL_main["ts"], np.array(want)), r
    assert np.array_equal(sols["fixed"].z, sols["upstream"].z)
    with pytest.raises(ValueError):
        O.amgb(O.fem1d(2), stop_rule="sometimes")


def test_decrement_centering_reaches_the_same_end_point():
    """VERDICT r2 item 5 (oracle CENTERING / mgb_amg_set_centering): following the path with the Newton-decrement rule and
    resolving only the last centre ends at the point the exact rule ends at, with fewer Newton steps on this mesh."""
    saved = O.CENTERING
    try:
        O.CENTERING = "exact"
        a = O.fem2d_solve(L=3, p=1.5)
        O.CENTERING = "decrement"
        b = O.fem2d_solve(L=3, p=1.5)
    finally:
        O.CENTERING = saved
    assert np.linalg.norm(a.z - b.z) <= 1e-11 * np.linalg.norm(a.z)
    assert np.array_equal(a.SOL_main["ts"], b.SOL_main["ts"])
    assert b.SOL_main["its"].sum() < a.SOL_main["its"].sum()


@pytest.mark.parametrize("kind,L,p", [(2, 3, 1.0), (2, 4, 1.5), (1, 4, 2.0), (3, 2, 1.0)])
def test_cpp_cpu_port_matches_the_numpy_oracle(kind, L, p):
    """oracle/cpu/mgb_cpu_newton.cpp (bench.py's native cpu_baseline) is pinned to the numpy oracle: same z
    (1e-10) and about the same Newton count on the default problems in 1-D, 2-D and 3-D."""
    import ctypes as C
    import __graft_entry__ as G
    lib = C.CDLL(G.build_cpu_port())
    dp, ip, lp = C.POINTER(C.c_double), C.POINTER(C.c_int), C.POINTER(C.c_longlong)
    lib.mgb_cpu_solve.argtypes = [C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int, dp, lp, dp, dp, ip]
    so = getattr(O, "fem%dd_solve" % kind)(L=L, p=p)
    zo = so.z.reshape(-1, order="F")
    z = np.zeros(zo.size)
    steps, sec, tr, th = C.c_longlong(), C.c_double(), C.c_double(), C.c_int()
    assert lib.mgb_cpu_solve(kind, L, 3, p, 0.0, 2, z.ctypes.data_as(dp), C.byref(steps), C.byref(sec), C.byref(tr), C.byref(th)) == 0
    assert np.linalg.norm(z - zo) <= 1e-10 * np.linalg.norm(zo)
    assert tr.value == so.SOL_main["ts"][-1] and th.value == 2
    assert abs(steps.value - int(so.SOL_main["its"].sum())) <= max(3, 0.25 * so.SOL_main["its"].sum())


def test_per_node_exponents_reduce_to_the_scalar_case():
    """an x-dependent exponent p(x) (upstream convex_Euclidian_power with a function p) enters as per-node a = 2 / p
    and mu(p); a constant p(x) must reproduce the scalar solve bit for bit, a varying one stays a valid convex problem."""
    g = O.fem2d(3)
    n = g.x.shape[0]
    a, b = O.amgb(g, p=1.5), O.amgb(g, p=np.full(n, 1.5))
    assert np.array_equal(a.z, b.z)
    c = O.amgb(g, p=lambda x: 1.5 + 0.4 * x[0])
    assert np.all(np.isfinite(c.z)) and np.abs(c.z - a.z).max() > 1e-3
    assert np.array_equal(O.barrier_mu(np.array([1.0, 2.0, 3.0])), np.array([1.0, 0.0, 2.0]))


def test_piecewise_set_reduces_to_the_intersection_when_everything_is_selected():
    """oracle ConvexPiecewise (upstream convex_piecewise, [UPSTREAM-UNVERIFIED]): with every piece selected everywhere it is the
    plain intersection, bit for bit; deselecting the obstacle on one half lets u drop below it there."""
    g = O.fem2d(3)
    f = lambda x: np.array([5.0, 0.0, 0.0, 1.0])
    gg = lambda x: np.array([0.3 + 0.5 * (x[0] ** 2 + x[1] ** 2), 100.0])
    ob = O.LinearBarrier([0], [1.0], -0.1)
    a = O.amgb(g, p=1.5, f=f, g=gg, extra=[ob])
    b = O.amgb(g, p=1.5, f=f, g=gg, extra=[ob], select=lambda x: (True, True))
    assert np.array_equal(a.z, b.z)
    c = O.amgb(g, p=1.5, f=f, g=gg, extra=[ob], select=lambda x: (True, x[0] > 0))
    assert c.z[g.x[:, 0] < 0, 0].min() < 0.1 < c.z[g.x[:, 0] > 0, 0].min()
