"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C ABI, against
the CPU oracle on identical seeded inputs, against the committed golden fixtures, and -- at sizes the
oracle cannot reach quickly -- through size-independent properties.

Tolerances (fp64): kernel-level quantities 1e-12 relative; solve-level z 1e-10 relative l2
(BASELINE.json north_star; docs/src/guide.md:186-188)."""
import glob
import os

import numpy as np
import pytest
import scipy.sparse as sp

import mgb_oracle as O

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
KTOL = 1e-12
ZTOL = 1e-10


@pytest.fixture(scope="module")
def M(gpu_required):
    import mgb_amd
    return mgb_amd


def rel(a, b):
    a, b = np.asarray(a, dtype=float), np.asarray(b, dtype=float)
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


# ---------------------------------------------------------------- hooks: the reference's own KATs
def test_hooks_known_answers(M):
    """test/test_helpers.jl:53-167 on device types."""
    A = M.amgb_zeros(M.HPCSparseMatrix(sp.identity(3)), 5, 5)
    assert isinstance(A, M.HPCSparseMatrix) and A.shape == (5, 5) and A.host.nnz == 0
    Bz = M.amgb_zeros(M.HPCMatrix(np.ones((2, 2))), 4, 6)
    assert isinstance(Bz, M.HPCMatrix) and Bz.shape == (4, 6) and not Bz.to_numpy().any()
    assert len(M.amgb_zeros(M.HPCVector, 7)) == 7
    assert M.amgb_all_isfinite(M.HPCVector([1.0, 2.0, 3.0])) is True
    assert M.amgb_all_isfinite(M.HPCVector([1.0, np.inf, 3.0])) is False
    assert M.amgb_all_isfinite(M.HPCVector([1.0, np.nan, 3.0])) is False
    assert M.amgb_all_isfinite(M.HPCMatrix(np.ones((3, 4)))) is True
    assert M.amgb_all_isfinite(M.HPCVector(np.zeros(0))) is True
    like = M.HPCSparseMatrix(sp.identity(3))
    D3 = M.amgb_diag(like, M.HPCVector([1.0, 2.0, 3.0]))
    assert D3.shape == (3, 3)
    D4 = M.amgb_diag(like, np.array([1.0, 2.0, 3.0, 4.0]))
    assert D4.shape == (4, 4)
    v = np.arange(1.0, 11.0)
    D10 = M.amgb_diag(like, M.HPCVector(v))
    assert (D10.to_scipy() != sp.diags(v)).nnz == 0               # test/test_diag.jl:28-46
    y = (D10 @ M.HPCVector(np.ones(10))).to_numpy()
    assert np.array_equal(y, v)
    C5 = M.amgb_blockdiag(M.HPCSparseMatrix(sp.identity(2)), M.HPCSparseMatrix(sp.identity(3)))
    assert C5.shape == (5, 5) and (C5.to_scipy() != sp.identity(5)).nnz == 0
    x = M.HPCMatrix(np.array([[1., 2], [3, 4], [5, 6]]))
    r = M.map_rows(lambda row: np.sum(row), x)
    assert isinstance(r, M.HPCVector) and r.to_numpy().tolist() == [3.0, 7.0, 11.0]
    x = M.HPCMatrix(np.array([[1., 2], [3, 4]]))
    r = M.map_rows(lambda row: np.array([np.sum(row), np.prod(row)]), x)
    assert isinstance(r, M.HPCMatrix) and r.to_numpy().tolist() == [[3.0, 2.0], [7.0, 12.0]]
    r = M.map_rows(lambda rx, ry: np.sum(rx) + ry[0], x, M.HPCVector([10.0, 20.0]))
    assert r.to_numpy().tolist() == [13.0, 27.0]


def test_map_rows_file_known_answers(M):
    """test/test_map_rows.jl:27-101."""
    v = np.arange(1.0, 9.0)
    hv, hw = M.HPCVector(v), M.HPCVector(v[::-1].copy())
    assert np.array_equal(M.map_rows(lambda x: x[0] ** 2, hv).to_numpy(), v ** 2)
    assert np.array_equal(M.map_rows(lambda x, y: x[0] * y[0], hv, hw).to_numpy(), v * v[::-1])
    assert np.array_equal(M.map_rows(lambda x: np.array([x[0], x[0] ** 2, x[0] ** 3]), hv).to_numpy(),
                          np.stack([v, v ** 2, v ** 3], 1))
    m = np.arange(1.0, 17.0).reshape(8, 2, order="F")
    assert np.array_equal(M.map_rows(lambda x: np.sum(x) ** 2, M.HPCMatrix(m)).to_numpy(), m.sum(1) ** 2)


# ---------------------------------------------------------------- SpMV / BLAS-1 kernels
@pytest.mark.parametrize("rows,cols,density", [(1, 1, 1.0), (3, 2, 1.0), (257, 63, 0.05), (1000, 1000, 0.002),
                                               (64, 5000, 0.3), (5000, 7, 0.9), (40, 40, 0.0)])
def test_spmv_matches_scipy(M, rows, cols, density):
    rng = np.random.default_rng(rows * 131 + cols)
    A = sp.random(rows, cols, density=density, random_state=rng, format="csr", data_rvs=rng.standard_normal)
    x = rng.standard_normal(cols)
    hA, hx = M.HPCSparseMatrix(A), M.HPCVector(x)
    y = (hA @ hx).to_numpy()
    want = A @ x
    assert np.abs(y - want).max() <= 1e-13 * max(1.0, np.abs(want).max()) * max(1, A.nnz // max(rows, 1))
    yt = (hA.T @ M.HPCVector(want)).to_numpy()                    # R' * v  (test/test_nonsquare.jl:62)
    wt = A.T @ want
    assert np.abs(yt - wt).max() <= 1e-12 * max(1.0, np.abs(wt).max())


def test_reference_sparse_algebra_kats(M):
    """test/test_basic_ops.jl:27-97 (A, B integer matrices; exact)."""
    A = M.HPCSparseMatrix(np.array([[1., 0], [2, 3], [0, 4]]))
    x = M.HPCVector([1.0, 1.0])
    assert (A @ x).to_numpy().tolist() == [1.0, 5.0, 4.0]
    assert (A.T @ M.HPCVector([1.0, 1.0, 1.0])).to_numpy().tolist() == [3.0, 7.0]
    # M*M, M'*M, M+M, hcat, blockdiag through the library's own sparse algebra (mgb_csr_spgemm / add / ...)
    An, Bn = np.array([[1., 0], [2, 3], [0, 4]]), np.array([[1., 2, 3], [4, 5, 6]])
    B = M.HPCSparseMatrix(Bn)
    AB = A @ B                                                    # test_basic_ops.jl:39
    assert isinstance(AB, M.HPCSparseMatrix) and AB.shape == (3, 3)
    assert np.array_equal(AB.to_scipy().toarray(), An @ Bn)
    AtA = A.T @ A                                                 # test_basic_ops.jl:55
    assert np.array_equal(AtA.to_scipy().toarray(), An.T @ An)
    assert np.array_equal((AB @ M.HPCVector([1.0, -1.0, 2.0])).to_numpy(), An @ Bn @ np.array([1.0, -1.0, 2.0]))
    S2 = AB + AB.T                                                # test_matrix_addition.jl:48-63
    assert np.array_equal(S2.to_scipy().toarray(), An @ Bn + (An @ Bn).T)
    assert (AB - AB).nnz == AB.nnz and not (AB - AB).to_scipy().toarray().any()      # structural: cancelled entries stay
    Hc = M.hcat(A, M.amgb_zeros(A, 3, 2), A)                      # test_d0_construction.jl:92-100
    assert np.array_equal(Hc.to_scipy().toarray(), np.hstack([An, np.zeros((3, 2)), An]))
    Bd = M.amgb_blockdiag(A, B)
    assert Bd.shape == (5, 5) and np.array_equal(Bd.to_scipy().toarray(), sp.block_diag([An, Bn]).toarray())
    # (A'A + 0.01 I) \ 1 to 1e-10 (test_basic_ops.jl:75-97): through the AMG-free sparse types only the product is checked
    reg = AtA + M.amgb_diag(A, np.full(2, 0.01))
    assert np.allclose(reg.to_scipy().toarray(), An.T @ An + 0.01 * np.eye(2), rtol=0, atol=1e-15)


def test_operator_transpose_and_weighted_products(M):
    """test/test_transpose_only.jl:45,85,129 (D_dx', hcat(D_dx, Z)' and D0' * spdiagm(w .* y) on fem1d L=2, 1e-12) and
    test/test_partitions.jl:104 (D' * spdiagm(w) * D on fem1d L=3, 1e-10), through the library's device sparse types."""
    g, gn = M.fem1d_mpi(2), M.fem1d(2)
    n = len(gn.w)
    Dn = sp.csr_matrix(gn.operators["dx"])
    D = g.operators["dx"]
    assert abs(D.T.to_scipy() - Dn.T).max() < 1e-12                                   # Test 1
    D0, D0n = M.hcat(D, M.amgb_zeros(D, n, n)), sp.hstack([Dn, sp.csr_matrix((n, n))]).tocsr()
    assert D0.shape == (n, 2 * n) and abs(D0.T.to_scipy() - D0n.T).max() < 1e-12      # Test 2
    foo = M.amgb_diag(D, g.w * M.HPCVector(np.full(n, 0.5)))
    tmp = D0.T @ foo                                                                  # Test 3
    assert tmp.shape == (2 * n, n) and abs(tmp.to_scipy() - D0n.T @ sp.diags(gn.w * 0.5)).max() < 1e-12
    g3, g3n = M.fem1d_mpi(3), M.fem1d(3)
    D3, D3n = g3.operators["dx"], sp.csr_matrix(g3n.operators["dx"])
    DtwD = D3.T @ (M.amgb_diag(D3, g3.w) @ D3)
    want = (D3n.T @ sp.diags(g3n.w) @ D3n).toarray()
    assert np.linalg.norm(DtwD.to_scipy().toarray() - want) < 1e-10


def test_vector_ops(M):
    rng = np.random.default_rng(5)
    for n in (1, 63, 64, 65, 100003):
        a, b = rng.standard_normal(n), rng.standard_normal(n)
        ha, hb = M.HPCVector(a), M.HPCVector(b)
        assert abs(ha.dot(hb) - np.dot(a, b)) <= 1e-12 * max(1.0, np.sqrt(n))
        assert np.array_equal((ha * hb).to_numpy(), a * b)       # w .* col (test_column_extract.jl:65)
        assert np.array_equal((ha + hb).to_numpy(), a + b)
        assert np.array_equal((ha - hb).to_numpy(), a - b)
        assert abs(ha.norm() - np.linalg.norm(a)) <= 1e-12 * max(1.0, np.sqrt(n))      # norm / sum (profile_scaling.jl:89-134)
        assert abs(ha.sum() - a.sum()) <= 1e-12 * max(1.0, np.sqrt(n))
    assert M.HPCVector(np.zeros(0)).sum() == 0.0 and M.HPCVector(np.zeros(0)).norm() == 0.0
    m = rng.standard_normal((50, 4))
    for j in range(4):                                                    # y[:,j] on the device (test_column_extract.jl:50)
        assert np.array_equal(M.HPCMatrix(m).column(j).to_numpy(), m[:, j])
    with pytest.raises(IndexError):
        M.HPCMatrix(m).column(4)
    wcol = M.HPCVector(np.arange(50.0)) * M.HPCMatrix(m).column(1)         # w .* y[:,1] (test_column_extract.jl:65)
    assert np.array_equal(wcol.to_numpy(), np.arange(50.0) * m[:, 1])


def test_nonsquare_restriction_fem1d(M):
    """test/test_nonsquare.jl:28-97: R is 16x7 at fem1d L=3; R*z, R'*v."""
    g = M.fem1d_mpi(3)
    R = g.subspaces["dirichlet"][-1]
    assert R.shape == (16, 7)
    z = np.sin(np.linspace(0, np.pi, 7))
    Rz = (R @ M.HPCVector(z)).to_numpy()
    assert np.abs(Rz - R.host @ z).max() < 1e-15
    v = np.cos(np.arange(16.0))
    assert np.abs((R.T @ M.HPCVector(v)).to_numpy() - R.host.T @ v).max() < 1e-14


# ---------------------------------------------------------------- barrier f0/f1/f2 vs oracle
def _problem(M, kind, L, p):
    gm = getattr(M, kind + "_mpi")(L)
    go = getattr(O, kind)(L)
    dim = go.discretization["dim"]
    A = M.AMG(gm, p=p)
    Mo = O.amg(go)
    x = Mo.x
    z0 = O.map_rows(lambda xi: O.DEFAULT_G[dim](xi), x).reshape(-1, order="F")
    c = O.map_rows(lambda xi: O.DEFAULT_F[dim](xi), x)
    A.set_c(c)
    A.set_z(z0)
    B = O.Barrier(O.convex_Euclidian_power(list(range(1, dim + 2)), p))
    return A, Mo, B, z0, c, go


def _match_columns(Ro, Rg):
    """permutation pi with Ro[:, j] == Rg[:, pi[j]] (continuous-dof numbering differs between the two
    independent geometry builders; broken-node quantities are compared directly)."""
    f = np.sin(np.arange(Ro.shape[0]) * 0.7 + 1.0)
    fo, fg = Ro.T @ f, Rg.T @ f
    po, pg = np.argsort(fo), np.argsort(fg)
    pi = np.empty(len(po), dtype=int)
    pi[po] = pg
    assert abs(Ro - Rg[:, pi]).max() < 1e-12
    return pi


@pytest.mark.parametrize("kind,L,p", [("fem1d", 3, 1.0), ("fem1d", 5, 1.5), ("fem2d", 2, 1.0), ("fem2d", 3, 1.5),
                                      ("fem2d", 3, 2.0), ("fem2d", 2, 3.0), ("fem2d", 4, 1.0), ("fem3d", 2, 1.5)])
def test_barrier_kernels_match_oracle_all_levels(M, kind, L, p):
    A, Mo, B, z0, c, go = _problem(M, kind, L, p)
    rng = np.random.default_rng(11)
    t = 3.7
    gm_sub = A.geometry.subspaces
    for l in range(L):
        Ro = Mo.R[l]
        Rg = sp.block_diag([gm_sub["dirichlet"][l].host, gm_sub["full"][l].host], format="csr")
        pi = _match_columns(Ro, Rg)
        N = Ro.shape[1]
        assert A.level_size(l)[0] == N
        so = 2e-3 * rng.standard_normal(N)
        sg = np.zeros(N)
        sg[pi] = so                                               # same function, product numbering
        Dz_o = B.apply_D(Mo.D, z0 + Ro @ so)
        assert rel(A.apply_D(l, sg), Dz_o) < KTOL                 # a3: apply_D (test_apply_d.jl:44)
        y_o = B.f0(so, Mo.x, Mo.w, t * c, Ro, Mo.D, z0)
        y_g, parts = A.f0(l, sg, t, parts=True)
        assert np.isfinite(y_o) and abs(y_g - y_o) <= KTOL * abs(y_o)     # a4
        g_o = B.f1(so, Mo.x, Mo.w, t * c, Ro, Mo.D, z0)
        g_g = A.f1(l, sg, t)
        assert rel(g_g[pi], g_o) < 1e-11                          # a5
        H_o = B.f2(so, Mo.x, Mo.w, t * c, Ro, Mo.D, z0).toarray()
        H_g, lower = A.f2(l, sg, t)
        H_g = H_g.toarray()[np.ix_(pi, pi)]
        assert np.abs(H_g - H_o).max() <= 1e-11 * np.abs(H_o).max()       # a6 (tolerance of test_matrix_addition.jl:86-95 scaled)
        assert np.abs(H_g - H_g.T).max() == 0                      # assembled from one triangle: exactly symmetric
        n_o = O.solve(sp.csr_matrix(H_o), g_o)                    # a7: solve(A,b) = A \ b
        for solver in ("gpu", "host"):
            n_g = A.solve_linear(l, lower, g_g, solver=solver)
            assert rel(n_g[pi], n_o) < 1e-9
            Hg, _ = A.f2(l, sg, t)
            assert np.linalg.norm(Hg @ n_g - g_g) <= 1e-10 * np.linalg.norm(g_g)     # small residual


@pytest.mark.parametrize("kind,L,p", [("fem1d", 5, 1.5), ("fem2d", 4, 1.0), ("fem2d", 3, 2.0), ("fem3d", 2, 1.5)])
def test_float32_kernels_match_the_oracle_and_the_double_kernels(M, kind, L, p):
    """SURVEY.md section 8 f3 (the reference runs Float32 on its Metal backend, test/test_utils.jl:67-88): the SpMV / barrier
    kernels instantiated for float (csrc/kernels_f32.hip) at every level -- f0, f1, f2 against the ORACLE (float64 numpy) at
    float32 accuracy, tolerance 2e-5 relative (eps_f32 = 1.2e-7 times the growth of the row sums and of 1/phi) -- and the
    double instantiation of the same kernel templates against the production double kernels bit for bit, which ties the
    float kernels to the code the solve runs."""
    A, Mo, B, z0, c, go = _problem(M, kind, L, p)
    rng = np.random.default_rng(23)
    t = 3.7
    gm_sub = A.geometry.subspaces
    for l in range(L):
        Ro = Mo.R[l]
        Rg = sp.block_diag([gm_sub["dirichlet"][l].host, gm_sub["full"][l].host], format="csr")
        pi = _match_columns(Ro, Rg)
        N = Ro.shape[1]
        so = 2e-3 * rng.standard_normal(N)
        sg = np.zeros(N)
        sg[pi] = so
        y_o = B.f0(so, Mo.x, Mo.w, t * c, Ro, Mo.D, z0)
        g_o = B.f1(so, Mo.x, Mo.w, t * c, Ro, Mo.D, z0)
        H_o = B.f2(so, Mo.x, Mo.w, t * c, Ro, Mo.D, z0).toarray()
        y32 = A.f0_f32(l, sg, t)
        g32 = A.f1_f32(l, sg, t)
        v32 = A.f2_f32(l, sg, t)
        assert g32.dtype == np.float32 and v32.dtype == np.float32
        assert abs(y32 - y_o) <= 2e-5 * abs(y_o)
        assert rel(g32[pi].astype(np.float64), g_o) < 2e-5
        rp, ci = A.hessian_pattern(l)
        Lo = sp.csr_matrix((v32.astype(np.float64), ci, rp), shape=(N, N))
        H32 = (Lo + sp.tril(Lo, -1).T).toarray()[np.ix_(pi, pi)]
        assert np.abs(H32 - H_o).max() <= 2e-5 * np.abs(H_o).max()
        # the same templates, T = double: the production kernels, bit for bit
        assert np.array_equal(A.f1_template_f64(l, sg, t), A.f1(l, sg, t))
        # f2: the template instantiation goes through the plan T, the production path assembles element by element (round 3):
        # the same recipe in another summation order
        assert rel(A.f2_template_f64(l, sg, t), A.f2(l, sg, t)[1]) < 1e-13
        # and float really is float: it differs from the double result, but only at float accuracy
        g64 = A.f1(l, sg, t)
        d = rel(g32.astype(np.float64), g64)
        assert 0 < d < 2e-5


def test_infeasible_trial_is_reported_not_raised(M):
    """amgb_all_isfinite semantics (src:121-133): an infeasible line-search trial is a status."""
    A, Mo, B, z0, c, go = _problem(M, "fem2d", 2, 1.0)
    l = 1
    N = A.level_size(l)[0]
    s = np.zeros(N)
    Rg = sp.block_diag([A.geometry.subspaces["dirichlet"][l].host, A.geometry.subspaces["full"][l].host])
    s[Rg.shape[1] - 5:] = -1000.0                                  # push the slack s far below |grad u|
    y = A.f0(l, s, 1.0)
    assert not np.isfinite(y)
    assert np.isfinite(A.f0(l, np.zeros(N), 1.0))


def test_fraction_to_boundary_trial_matches_oracle(M):
    """Line-search trial semantics: finite only if every row keeps >= FRAC_TO_BOUNDARY of its cone distance."""
    A, Mo, B, z0, c, go = _problem(M, "fem2d", 3, 1.0)
    l = 2
    Ro = Mo.R[l]
    Rg = sp.block_diag([A.geometry.subspaces["dirichlet"][l].host, A.geometry.subspaces["full"][l].host], format="csr")
    pi = _match_columns(Ro, Rg)
    N = Ro.shape[1]
    rng = np.random.default_rng(2)
    d = rng.standard_normal(N)
    t = 1.0
    _, phi0 = B.f0_phi(np.zeros(N), Mo.x, Mo.w, t * c, Ro, Mo.D, z0)
    seen = set()
    for alpha in (1e-4, 1e-2, 0.3, 1.0, 3.0, 10.0, 30.0):
        so = alpha * d
        sg = np.zeros(N)
        sg[pi] = so
        yo, _ = B.f0_phi(so, Mo.x, Mo.w, t * c, Ro, Mo.D, z0, phi_ref=phi0)
        yg = A.f0_trial(l, np.zeros(N), sg, t)
        assert np.isfinite(yo) == np.isfinite(yg)
        seen.add(bool(np.isfinite(yo)))
        if np.isfinite(yo):
            assert abs(yg - yo) <= KTOL * abs(yo)
    assert seen == {True, False}                                  # the sweep crosses the rule


# ---------------------------------------------------------------- barrier menu: power cone intersected with a half space
OBST_G = {2: lambda x: np.array([0.3 + 0.5 * (x[0] ** 2 + x[1] ** 2), 100.0]), 1: lambda x: np.array([0.3 + 0.5 * x[0] ** 2, 100.0])}
OBST_F = {2: lambda x: np.array([5.0, 0.0, 0.0, 1.0]), 1: lambda x: np.array([5.0, 0.0, 1.0])}


@pytest.mark.parametrize("kind,L,p,psi", [("fem2d", 3, 1.0, 0.1), ("fem2d", 3, 2.0, 0.2), ("fem1d", 4, 1.5, 0.1)])
def test_obstacle_barrier_matches_oracle(M, kind, L, p, psi):
    """General barrier menu: the p-Laplace power cone intersected with the half space u - psi > 0 (upstream
    `intersect(convex_Euclidian_power, convex_linear)`: a constant lower obstacle).  With the load f = (5, 0.., 1) the
    unconstrained minimiser dips below psi (for p = 1 it is unbounded below), so the obstacle is active.  Kernel level (f0,
    f1, f2 at a random point) at 1e-12 / 1e-11 and the whole solve at 1e-10 against the oracle."""
    dim = 1 if kind == "fem1d" else 2
    gm = getattr(M, kind + "_mpi")(L)
    go = getattr(O, kind)(L)
    cone = (list(range(1, dim + 2)), p)
    lin = ("linear", [0], [1.0], -psi)
    A = M.AMG(gm, p=p, cones=[cone, lin])
    assert A.nY == (dim + 1) * (dim + 2) // 2 + 1
    Mo = O.amg(go)
    z0 = O.map_rows(lambda xi: OBST_G[dim](xi), Mo.x).reshape(-1, order="F")
    c = O.map_rows(lambda xi: OBST_F[dim](xi), Mo.x)
    A.set_c(c)
    A.set_z(z0)
    Bo = O.Barrier(O.ConeIntersection([O.convex_Euclidian_power(list(range(1, dim + 2)), p), O.LinearBarrier([0], [1.0], -psi)]))
    l = L - 1
    Ro = Mo.R[l]
    Rg = sp.block_diag([gm.subspaces["dirichlet"][l].host, gm.subspaces["full"][l].host], format="csr")
    pi = _match_columns(Ro, Rg)
    N = Ro.shape[1]
    rng = np.random.default_rng(3)
    so = 2e-3 * rng.standard_normal(N)
    sg = np.zeros(N)
    sg[pi] = so
    t = 2.5
    y_o = Bo.f0(so, Mo.x, Mo.w, t * c, Ro, Mo.D, z0)
    assert np.isfinite(y_o) and abs(A.f0(l, sg, t) - y_o) <= KTOL * abs(y_o)
    assert rel(A.f1(l, sg, t)[pi], Bo.f1(so, Mo.x, Mo.w, t * c, Ro, Mo.D, z0)) < 1e-11
    H_o = Bo.f2(so, Mo.x, Mo.w, t * c, Ro, Mo.D, z0).toarray()
    H_g = A.f2(l, sg, t)[0].toarray()[np.ix_(pi, pi)]
    assert np.abs(H_g - H_o).max() <= 1e-11 * np.abs(H_o).max()
    # a trial that crosses the obstacle is reported as infeasible, not raised
    sbad = np.zeros(N)
    sbad[: Rg.shape[1] // 3] = -10.0
    assert not np.isfinite(A.f0(l, sbad, t))
    sol = M.amgb(gm, p=p, f=OBST_F[dim], g=OBST_G[dim], cones=[cone, lin])
    ref = O.amgb(go, p=p, f=OBST_F[dim], g=OBST_G[dim], extra=[O.LinearBarrier([0], [1.0], -psi)])
    z = M.mpi_to_native(sol).z
    assert rel(z, ref.z) < ZTOL
    assert z[:, 0].min() > psi and z[:, 0].min() - psi < 1e-3            # strictly feasible, and the obstacle is active
    with pytest.raises(M.MGBError):                                          # bad term descriptions are argument errors
        M.AMG(gm, p=p, cones=[cone, ("linear", [0, 0], [1.0, 1.0], 0.0)])


# ---------------------------------------------------------------- whole solves
CASES = [("fem1d", 3, 1.0), ("fem1d", 4, 2.0), ("fem2d", 2, 1.5), ("fem2d", 3, 1.0), ("fem2d", 3, 2.0),
         ("fem3d", 2, 1.0), ("fem3d", 2, 2.0)]


@pytest.mark.parametrize("kind,L,p", CASES)
def test_solve_matches_oracle_and_golden(M, kind, L, p):
    sol = getattr(M, kind + "_mpi_solve")(L=L, p=p, verbose=False)
    native = M.mpi_to_native(sol)
    z = native.z
    assert isinstance(sol.z, M.HPCMatrix) and isinstance(z, np.ndarray)          # test_2d.jl:144
    gold = np.load(os.path.join(HERE, "golden", "%s_L%d_p%s.npz" % (kind, L, str(p).replace(".", "_"))))
    assert z.shape == gold["z"].shape
    assert rel(z, gold["z"]) < ZTOL
    assert np.abs(z - gold["z"]).max() < 1e-9                                     # guide.md:246-253 reports sup-norm diffs <= 4e-11
    assert np.allclose(sol.SOL_main["ts"], gold["ts"], rtol=1e-12)
    # intermediate centres are only converged to the Newton stopping rule; the final objective is tight
    assert rel(sol.SOL_main["c_dot_Dz"], gold["c_dot_Dz"]) < 1e-6
    assert abs(sol.SOL_main["c_dot_Dz"][-1] - gold["c_dot_Dz"][-1]) <= 1e-9 * abs(gold["c_dot_Dz"][-1])
    its, gits = sol.SOL_main["its"], gold["its"]
    assert its.shape == gits.shape
    # the exact stopping rule compares rounding-level quantities (y_next >= y_min, |g| ratios), so the last
    # one or two Newton steps of each centering may differ between implementations; z is what is pinned
    assert abs(int(its.sum()) - int(gits.sum())) <= max(3, 0.25 * gits.sum())
    # live oracle run on the same inputs (the differential check the reference's CI does at run time,
    # test/test_quick.jl:137-140 with 1e-7; we hold 1e-10)
    zo = getattr(O, kind + "_solve")(L=L, p=p).z
    assert rel(z, zo) < ZTOL


@pytest.mark.parametrize("kind,L,p", [("fem1d", 4, 1.0), ("fem2d", 3, 1.5), ("fem2d", 5, 1.5)])
def test_level_loop_schedule_matches_oracle(M, kind, L, p):
    """The literal coarse -> fine level loop (schedule='all', SURVEY §3.1 amgb_step) against the oracle run
    with the same schedule; it must also land on the same z as the default schedule.  fem2d L=5 is a mesh whose
    fine levels have multi-panel fronts (oracle: 652 Newton steps against 96 with the default schedule,
    profiles/r2_schedule_newton_counts.txt)."""
    sol = getattr(M, kind + "_mpi_solve")(L=L, p=p, schedule="all")
    z = M.mpi_to_native(sol).z
    so = getattr(O, kind + "_solve")(L=L, p=p, schedule="all")
    assert rel(z, so.z) < ZTOL
    its = sol.SOL_main["its"]
    assert its.shape[0] == L and np.all(its.sum(axis=1) > 0)                  # every level visited
    assert abs(int(its.sum()) - int(so.SOL_main["its"].sum())) <= max(3, 0.25 * so.SOL_main["its"].sum())
    zf = M.mpi_to_native(getattr(M, kind + "_mpi_solve")(L=L, p=p)).z
    assert rel(z, zf) < 1e-9


def test_device_cholesky_matches_host_on_large_level(M):
    """GpuChol vs MfChol vs the assembled matrix at fem2d L=6 (12 290 unknowns, ~700 fronts)."""
    A, Mo, B, z0, c, go = _problem(M, "fem2d", 6, 1.0)
    l = 5
    N = A.level_size(l)[0]
    H, lower = A.f2(l, np.zeros(N), 1.0)
    g = A.f1(l, np.zeros(N), 1.0)
    xg = A.solve_linear(l, lower, g, solver="gpu")
    xh = A.solve_linear(l, lower, g, solver="host")
    assert rel(xg, xh) < 1e-10
    assert np.linalg.norm(H @ xg - g) <= 1e-10 * np.linalg.norm(g)
    assert np.array_equal(xg, A.solve_linear(l, lower, g, solver="gpu"))      # bitwise reproducible
    bad = lower.copy()
    rp, ci = A.hessian_pattern(l)
    diag = np.flatnonzero(ci == np.repeat(np.arange(N), np.diff(rp)))
    bad[diag[N // 2]] = -1.0                                                    # indefinite -> status, not a crash
    with pytest.raises(M.MGBError) as ei:
        A.solve_linear(l, bad, g, solver="gpu")
    assert ei.value.code == -3


def test_host_and_device_solver_paths_agree(M):
    zg = M.mpi_to_native(M.fem2d_mpi_solve(L=3, p=1.0, solver="gpu")).z
    zh = M.mpi_to_native(M.fem2d_mpi_solve(L=3, p=1.0, solver="host")).z
    assert rel(zg, zh) < ZTOL


@pytest.mark.parametrize("kind,L,p", [("fem2d", 5, 1.5), ("fem1d", 10, 1.0), ("fem3d", 3, 1.0)])
def test_solve_properties_at_scale(M, kind, L, p):
    """Size-independent properties where a live oracle run would be slow (BASELINE configs[1])."""
    sol = getattr(M, kind + "_mpi_solve")(L=L, p=p)
    z = M.mpi_to_native(sol).z
    g = M.mpi_to_native(sol.geometry)
    dim = g.discretization["dim"]
    x = g.x
    gfun = O.DEFAULT_G[dim]
    full, dirichlet = g.subspaces["full"][-1], g.subspaces["dirichlet"][-1]
    bnd = np.asarray((full @ np.ones(full.shape[1])) - (dirichlet @ np.ones(dirichlet.shape[1]))).ravel() > 0.5
    u0 = np.array([gfun(xi)[0] for xi in x])
    assert np.abs(z[bnd, 0] - u0[bnd]).max() < 1e-13                      # Dirichlet data untouched
    grad2 = sum((g.operators[k] @ z[:, 0]) ** 2 for k in ("dx", "dy", "dz")[:dim])
    assert np.all(z[:, 1] > grad2 ** (p / 2))                               # strictly inside the cone
    cd = sol.SOL_main["c_dot_Dz"]
    assert np.all(np.diff(cd) <= 1e-9 * abs(cd[0]))                         # objective decreases along the path
    assert sol.SOL_main["ts"][-1] > 1 / np.sqrt(np.finfo(float).eps)
    # u is continuous: broken values agree on shared nodes  (z = z0 + R s with R continuous)
    proj = full @ sp.linalg.lsqr(full, z[:, 0] - 0.0, atol=1e-15, btol=1e-15)[0]
    assert np.abs(proj - z[:, 0]).max() < 1e-8
    # duality-gap style bound: barrier parameter of the cone barrier is <= 4 per node
    assert 0 < cd[-2] - cd[-1] < 4 * np.sum(g.w) / sol.SOL_main["ts"][-2] * 1.01 + 1e-12


def test_native_to_mpi_roundtrip(M):
    """test/test_quick.jl:91-94 / examples/roundtrip_conversion.jl: geometry round trip at 1e-10."""
    g = M.fem2d(2)
    gm = M.native_to_mpi(g)
    assert isinstance(gm.x, M.HPCMatrix) and isinstance(gm.w, M.HPCVector)
    assert all(isinstance(v, M.HPCSparseMatrix) for v in gm.operators.values())
    back = M.mpi_to_native(gm)
    assert np.allclose(back.x, g.x, atol=1e-10) and np.allclose(back.w, g.w, atol=1e-10)
    for k in g.operators:
        assert abs(back.operators[k] - g.operators[k]).max() < 1e-10
    for k in g.subspaces:
        for a, b in zip(back.subspaces[k], g.subspaces[k]):
            assert abs(a - b).max() < 1e-10
    with pytest.raises(ValueError):
        M.native_to_mpi(g, Ti=np.int64)


LARGE_CASES = [("fem2d", 7, 1.0), ("fem2d", 7, 1.5), ("fem3d", 4, 1.0)]      # BASELINE.json configs[2], configs[3]


def test_all_golden_files_are_covered():
    have = {os.path.basename(f) for f in glob.glob(os.path.join(HERE, "golden", "*.npz"))}
    want = {"%s_L%d_p%s.npz" % (k, L, str(p).replace(".", "_")) for k, L, p in CASES}
    want |= {"large_%s_L%d_p%s.npz" % (k, L, str(p).replace(".", "_")) for k, L, p in LARGE_CASES}
    want |= {"large_parabolic_L6_p1_0.npz"}
    assert have - {"large_fem2d_L8_p1_0.npz"} == want        # (the L=8 fixture is optional: test_beyond_the_headline_size_fem2d_L8)


# Newton on the finest level stops on stagnation of the objective (oracle stopping_exact), which resolves the centre of the last
# t only up to the last step NOT taken; whether that step is taken hangs on the rounding of an objective of size 1e9.  In 2-D
# both sides end within 3e-13 of the exact centre; in 3-D (fem3d L=4) the oracle stops 2.0e-10 short (1.2e-11 in u, 2.9e-10 in
# the slack) and the HIP path 1.9e-13 (tests/golden/polish_centre.py, profiles/r2_fem3d_L4_centre.txt).  So the end points are
# compared at the stop rule's resolution, and the HIP end point at 1e-11 with the oracle's EXACT centre `z_centre` (the oracle end
# point polished by Newton steps whose gradient is evaluated in 80-bit extended precision -- oracle code only).
LARGE_END_POINT_TOL = {"fem2d": ZTOL, "fem3d": 5e-10}
LARGE_CENTRE_TOL = 1e-11


@pytest.mark.parametrize("kind,L,p", LARGE_CASES)
def test_headline_sizes_match_oracle_goldens(M, kind, L, p):
    """BASELINE.json configs at full size -- fem2d L=7 (the bench workload, p = 1 and 1.5) and fem3d L=4 -- against
    z of the CPU oracle on the same mesh (tests/golden/make_golden_large.py; minutes to hours of host time per case, so the
    GPU box reads the committed vectors).  Bar: relative l2 <= 1e-10 (BASELINE.json north_star) against the oracle's exact
    centre of t_final for every case, and against the oracle's own end point wherever its stop rule resolves that well (2-D);
    the reference's own two implementations differ by 3.3e-13 in the sup norm at L=7 (docs/src/guide.md:252).
    Measured to the end point / to the centre: fem2d p = 1 7.0e-13 / see log (466 Newton steps here against 527 in the oracle:
    different kappa histories, same end point), p = 1.5 2.5e-13, fem3d 2.0e-10 / 1.9e-13."""
    gold = np.load(os.path.join(HERE, "golden", "large_%s_L%d_p%s.npz" % (kind, L, str(p).replace(".", "_"))))
    sol = getattr(M, kind + "_mpi_solve")(L=L, p=p)
    z = M.mpi_to_native(sol).z
    assert z.shape == gold["z"].shape
    err, err_centre = rel(z, gold["z"]), rel(z, gold["z_centre"])
    print("%s L=%d p=%g: rel l2 to the oracle end point %.3e (u %.3e), to the exact centre %.3e (oracle itself %.3e)  newton %d "
          "(oracle %d)" % (kind, L, p, err, rel(z[:, 0], gold["z"][:, 0]), err_centre, float(gold["oracle_end_point_to_centre"]),
                           int(sol.SOL_main["its"].sum()), int(gold["its"].sum())))
    assert err_centre < LARGE_CENTRE_TOL
    assert err < LARGE_END_POINT_TOL[kind]
    assert rel(z[:, 0], gold["z"][:, 0]) < ZTOL                  # the solution component u: 1e-10 in every case
    assert np.allclose(sol.SOL_main["ts"][-1], gold["ts"][-1], rtol=1e-12)
    assert abs(sol.SOL_main["c_dot_Dz"][-1] - gold["c_dot_Dz"][-1]) <= 1e-9 * abs(gold["c_dot_Dz"][-1])


L8_GOLD = os.path.join(HERE, "golden", "large_fem2d_L8_p1_0.npz")


@pytest.mark.skipif(not os.path.exists(L8_GOLD), reason="fem2d L=8 golden not generated (the oracle needs hours at this size)")
def test_beyond_the_headline_size_fem2d_L8(M):
    """One size beyond BASELINE's headline: fem2d L=8, p = 1 (229 376 rows, 668 Newton steps on the device, 1.3 s) against the
    oracle's vector.  The fixture holds every 4th row plus functionals of the full vectors (column norms, a fixed projection:
    tests/golden/subsample_golden.py), so the whole solution is checked through size-independent numbers."""
    gold = np.load(L8_GOLD)
    sol = M.fem2d_mpi_solve(L=8, p=1.0)
    z = M.mpi_to_native(sol).z
    n, st = int(gold["n_full"]), int(gold["stride"])
    assert z.shape == (n, 2)
    probe = np.sin(0.37 * np.arange(n) + 0.11)
    for key, tol in (("z_centre", LARGE_CENTRE_TOL), ("z", ZTOL)):
        if key not in gold.files:
            continue
        assert rel(z[::st], gold[key]) < tol
        assert np.all(np.abs(np.linalg.norm(z, axis=0) - gold[key + "_colnorm"]) <= tol * gold[key + "_colnorm"])
        assert np.all(np.abs(probe @ z - gold[key + "_probe"]) <= tol * np.linalg.norm(probe) * gold[key + "_colnorm"])
    assert np.allclose(sol.SOL_main["ts"][-1], gold["ts"][-1], rtol=1e-12)


def test_golden_level_vectors(M):
    """Per-level f0/f1/f2 fingerprints stored with each golden solve."""
    for kind, L, p in CASES:
        gold = np.load(os.path.join(HERE, "golden", "%s_L%d_p%s.npz" % (kind, L, str(p).replace(".", "_"))))
        A, Mo, B, z0, c, go = _problem(M, kind, L, p)
        for l in range(L):
            Ro = Mo.R[l]
            Rg = sp.block_diag([A.geometry.subspaces["dirichlet"][l].host, A.geometry.subspaces["full"][l].host],
                               format="csr")
            pi = _match_columns(Ro, Rg)
            sg = np.zeros(Ro.shape[1])
            sg[pi] = gold["s_%d" % l]
            assert abs(A.f0(l, sg, 2.5) - gold["f0_%d" % l]) <= KTOL * abs(gold["f0_%d" % l])
            assert rel(A.f1(l, sg, 2.5)[pi], gold["f1_%d" % l]) < 1e-11
            H, _ = A.f2(l, sg, 2.5)
            assert rel(H.diagonal()[pi], gold["f2diag_%d" % l]) < 1e-11
            assert abs(np.sqrt(H.multiply(H).sum()) - gold["f2fro_%d" % l]) <= 1e-11 * gold["f2fro_%d" % l]


def test_prepare_moves_the_factorisation_build_out_of_the_solve(M):
    """AMG.prepare() (mgb_amg_prepare) builds level + factorisation structures up front; the solve that follows
    gives the same answer as a lazily built one."""
    g = M.fem2d_mpi(3)
    x = g.x.to_numpy()
    c = np.vstack([M.DEFAULT_F[2](xi) for xi in x])
    z0 = np.vstack([M.DEFAULT_G[2](xi) for xi in x]).reshape(-1, order="F")
    out = []
    for prep in (True, False):
        A = M.AMG(g, p=1.5)
        A.set_c(c)
        A.set_z(z0)
        if prep:
            A.prepare()
        A.solve()
        out.append(A.get_z())
    assert np.array_equal(out[0], out[1])


def test_more_than_eight_rows_of_D_are_rejected(M):
    g = M.fem1d_mpi(3)
    D = (("u", "id"),) * 8 + (("s", "id"),)
    with pytest.raises(M.MGBError):
        M.AMG(g, D=D, idx=[7, 8])


def test_quick_jl_custom_tolerance(M):
    """test/test_quick.jl:108,137-140: fem1d L=3, `amgb(g; p=1, tol=1e-10)` (a non-default tolerance: the
    t-continuation runs to 1/tol = 1e10), MPI-vs-native difference < 1e-7 there; here device vs oracle at 1e-10."""
    sol = M.fem1d_mpi_solve(L=3, p=1.0, tol=1e-10)
    ref = O.fem1d_solve(L=3, p=1.0, tol=1e-10)
    z = M.mpi_to_native(sol).z
    assert sol.SOL_main["ts"][-1] > 1e10 and ref.SOL_main["ts"][-1] > 1e10
    assert np.linalg.norm(z - ref.z) < 1e-7
    assert np.linalg.norm(z - ref.z) <= 1e-10 * np.linalg.norm(ref.z)


@pytest.mark.parametrize("kind,L,p,kw", [("fem1d", 1, 1.0, {}), ("fem2d", 1, 3.0, {}), ("fem3d", 1, 2.0, {"k": 2}),
                                         ("fem3d", 2, 1.5, {"k": 1})])
def test_tiny_meshes_match_oracle(M, kind, L, p, kw):
    """Smallest meshes (4 .. 64 rows): single-front elimination trees, empty boundaries, fronts below one panel."""
    sol = getattr(M, kind + "_mpi_solve")(L=L, p=p, **kw)
    ref = getattr(O, kind + "_solve")(L=L, p=p, **kw)
    z = M.mpi_to_native(sol).z
    assert z.shape == ref.z.shape
    assert np.linalg.norm(z - ref.z) <= 1e-10 * np.linalg.norm(ref.z)


def test_newton_matrix_capture_in_reference_layout(M):
    """SURVEY 8(f)4 / test/test_newton_matrix_compare.jl:33-51: the assembled Newton matrix leaves the library as an
    HPCSparseMatrix whose per-rank blocks (src:216-221) stack back to the same matrix, and A \\ b on it agrees."""
    A, Mo, B, z0, c, go = _problem(M, "fem2d", 3, 1.5)
    l = 2
    N = A.level_size(l)[0]
    s = np.zeros(N)
    H, lower = A.f2(l, s, 1.0)
    Hm = A.f2_hpc(l, s, 1.0)
    assert Hm.shape == (N, N) and abs(Hm.to_scipy() - H).max() == 0
    blocks = [Hm.local_block(r, 4) for r in range(4)]
    assert blocks[0]["row_partition"][-1] == N + 1 and blocks[0]["colptr"].dtype == np.int32
    back = M.HPCSparseMatrix.from_local_blocks(blocks)
    assert abs(back.to_scipy() - H).max() == 0
    g = A.f1(l, s, 1.0)
    x = A.solve_linear(l, lower, g)
    assert rel((back @ M.HPCVector(x)).to_numpy(), g) < 1e-10


def test_fused_and_separate_objective_kernels_agree(M, monkeypatch):
    """The line search evaluates f0 with one fused launch on launch-bound meshes and with waxpby + apply_D + barrier_f0
    beyond (csrc/amg.cpp: enqueue_f0); Dz is bitwise the same, the sums differ in summation order only."""
    zf = M.mpi_to_native(M.fem2d_mpi_solve(L=4, p=1.0)).z
    monkeypatch.setenv("MGB_FUSED_TRIAL_ROWS", "0")
    sol = M.fem2d_mpi_solve(L=4, p=1.0)
    zs = M.mpi_to_native(sol).z
    assert rel(zs, zf) < ZTOL
    A = M.AMG(M.fem2d_mpi(3), p=1.5)
    go = O.fem2d(3)
    Mo = O.amg(go)
    z0 = O.map_rows(lambda xi: O.DEFAULT_G[2](xi), Mo.x).reshape(-1, order="F")
    c = O.map_rows(lambda xi: O.DEFAULT_F[2](xi), Mo.x)
    A.set_c(c)
    A.set_z(z0)
    s = np.zeros(A.level_size(2)[0])
    y_sep = A.f0(2, s, 2.0)
    monkeypatch.delenv("MGB_FUSED_TRIAL_ROWS")
    B = M.AMG(M.fem2d_mpi(3), p=1.5)
    B.set_c(c)
    B.set_z(z0)
    y_fused = B.f0(2, s, 2.0)
    assert abs(y_sep - y_fused) <= 1e-13 * abs(y_fused)
    assert np.array_equal(A.apply_D(2, s), B.apply_D(2, s))


@pytest.mark.parametrize("kind,L,p,tol", [("fem2d", 5, 1.5, ZTOL), ("fem2d", 6, 2.0, ZTOL), ("fem2d", 5, 1.0, ZTOL)])
def test_solve_matches_live_oracle_on_multi_panel_meshes(M, kind, L, p, tol):
    """Meshes whose elimination trees have multi-panel fronts (front_step, the matrix-core updates, the backward
    split) against a live oracle run (2-20 s of host time).  Measured: 1e-14 (L=5, p=1.5), 1.3e-13 (L=6, p=2); the p = 1
    case (total variation, the ill-conditioned one) is back at the common 1e-10 bound since the continuation ends at a fixed
    t (round 1 needed 5e-10 there: the two runs could end at different t)."""
    zo = getattr(O, kind + "_solve")(L=L, p=p).z
    sol = getattr(M, kind + "_mpi_solve")(L=L, p=p)
    z = M.mpi_to_native(sol).z
    assert rel(z, zo) < tol


@pytest.mark.parametrize("L,p", [(3, 1.5), (4, 1.0), (4, 2.0)])
def test_x_dependent_obstacle_matches_oracle(M, L, p):
    """Barrier menu: an obstacle that varies in space, u > psi(x).  psi enters as a state variable WITHOUT
    unknowns (subspace "fixed": an n x 0 block of R) read through one more row `psi id` of D, so the half space
    1 u - 1 psi > 0 is the constant-coefficient term the kernels already have; the solve never moves psi.  Load 5 pushes u
    from its boundary value 1 onto the paraboloid 0.6 - 2 |x|^2: the contact set is non-trivial.  z against the oracle."""
    state = (("u", "dirichlet"), ("s", "full"), ("psi", "fixed"))
    D = (("u", "id"), ("u", "dx"), ("u", "dy"), ("s", "id"), ("psi", "id"))
    psi = lambda x: 0.6 - 2.0 * float(x[0] ** 2 + x[1] ** 2)
    f = lambda x: np.array([5.0, 0.0, 0.0, 1.0, 0.0])
    g = lambda x: np.array([1.0, 10.0, psi(x)])
    gm, go = M.fem2d_mpi(L), O.fem2d(L)
    sol = M.amgb(gm, p=p, state_variables=state, D=D, f=f, g=g, cones=[([1, 2, 3], p), ("linear", [0, 4], [1.0, -1.0], 0.0)])
    ref = O.amgb(go, p=p, state_variables=state, D=D, f=f, g=g, extra=[O.LinearBarrier([0, 4], [1.0, -1.0], 0.0)],
                 cone_idx=[1, 2, 3])
    z = M.mpi_to_native(sol).z
    assert z.shape == ref.z.shape == (go.x.shape[0], 3)
    assert np.array_equal(z[:, 2], np.array([psi(x) for x in go.x]))          # the obstacle is data: untouched, bit for bit
    assert rel(z[:, 0], ref.z[:, 0]) < ZTOL and rel(z, ref.z) < 1e-9          # (slack column: stop-rule resolution, as above)
    gap = z[:, 0] - z[:, 2]
    assert gap.min() > 0 and gap.min() < 1e-6 and (gap < 1e-4).sum() > 5       # strictly above, and in contact on a set of nodes


@pytest.mark.parametrize("kind,L,p", [("fem1d", 4, 1.5), ("fem2d", 3, 1.5), ("fem2d", 4, 1.0)])
def test_general_feasibility_phase_matches_oracle(M, kind, L, p):
    """SOL_feasibility (src:428-455) beyond the closed-form shift: the start u0 dips below the obstacle u > -0.2 inside the
    domain AND its cone slack is too small, so neither set is satisfied.  Both sides run the same phase 1 -- the problem
    relaxed by a slack field with a big-M penalty, path-followed until the slack is negative everywhere (oracle
    amgb_phase1_slack; device: one more state variable and D row, mgb_amg_set_early_stop) -- and then the main phase:
    same number of phase-1 centerings, z of the whole solve at 1e-10."""
    dim = 1 if kind == "fem1d" else 2
    f = (lambda x: np.array([0.5, 0.0, 1.0])) if dim == 1 else (lambda x: np.array([0.5, 0.0, 0.0, 1.0]))
    g = lambda x: np.array([1.0 - 1.5 * (1.0 - float(np.sum(np.asarray(x) ** 2)) / dim), 0.05])
    gm = getattr(M, kind + "_mpi")(L)
    go = getattr(O, kind)(L)
    cone = (list(range(1, dim + 2)), p)
    lin = ("linear", [0], [1.0], 0.2)
    u0 = np.array([g(xi)[0] for xi in go.x])
    assert u0.min() < -0.2                                                   # the start violates the obstacle
    sol = M.amgb(gm, p=p, f=f, g=g, cones=[cone, lin])
    ref = O.amgb(go, p=p, f=f, g=g, extra=[O.LinearBarrier([0], [1.0], 0.2)])
    Fg, Fo = sol.SOL_feasibility, ref.SOL_feasibility
    assert Fg is not None and Fo is not None
    assert abs(Fg["sigma0"] - Fo["sigma0"]) <= 1e-12 * Fo["sigma0"]
    assert np.allclose(Fg["ts"], Fo["ts"], rtol=1e-12) and len(Fg["ts"]) >= 2           # stopped early, on the same centering
    z = M.mpi_to_native(sol).z
    # u at 1e-10; the cone slack s (1e-4 .. 3 here) carries the resolution of the Newton stop rule (see LARGE_END_POINT_TOL):
    # 1.2e-10 measured on fem2d L=3, the two main phases start from phase-1 points that differ in the last bits
    assert rel(z[:, 0], ref.z[:, 0]) < ZTOL and rel(z, ref.z) < 1e-9
    assert z[:, 0].min() > -0.2
    # an infeasible PROBLEM (obstacle above the Dirichlet data on the boundary) is reported, not looped on
    with pytest.raises(M.MGBError):
        M.amgb(gm, p=p, f=f, g=g, cones=[cone, ("linear", [0], [1.0], -5.0)])


def test_map_rows_recognises_the_barrier_family(M):
    """`map_rows(F | F1 | F2, x, Dz)` with the row functions of a convex set runs on the device
    (mgb_map_rows_barrier: the fused kernels of the Newton path with unit weights) and equals the oracle's row maps; any other
    closure takes the host fallback and gives the same numbers for the same function."""
    g = M.fem2d_mpi(3)
    go = O.fem2d(3)
    Mo = O.amg(go)
    rng = np.random.default_rng(4)
    n = Mo.x.shape[0]
    z0 = O.map_rows(lambda xi: O.DEFAULT_G[2](xi), Mo.x).reshape(-1, order="F")
    Dz = O.Barrier.apply_D(Mo.D, z0) + 1e-3 * rng.standard_normal((n, 4))
    hDz = M.HPCMatrix(Dz)
    for cones, Q in (([([1, 2, 3], 1.5)], O.convex_Euclidian_power([1, 2, 3], 1.5)),
                     ([([1, 2, 3], 1.0), ("linear", [0], [1.0], 5.0)],
                      O.ConeIntersection([O.convex_Euclidian_power([1, 2, 3], 1.0), O.LinearBarrier([0], [1.0], 5.0)]))):
        F, F1, F2 = M.barrier_functions(cones, 4)
        y0 = M.map_rows(F, g.x, hDz)
        assert isinstance(y0, M.HPCVector) and rel(y0.to_numpy(), Q.F(Mo.x, Dz)) < KTOL
        y1 = M.map_rows(F1, g.x, hDz)
        assert isinstance(y1, M.HPCMatrix) and y1.shape == (n, 4) and rel(y1.to_numpy(), Q.F1(Mo.x, Dz)) < KTOL
        y2 = M.map_rows(F2, g.x, hDz)
        assert y2.shape == (n, 16) and rel(y2.to_numpy(), Q.F2(Mo.x, Dz).reshape(n, 16)) < KTOL
        # w .* y[:, jk] as the Hessian recipe consumes it (test_column_extract.jl:50-80), all on the device
        col = (g.w * y2.column(1 * 4 + 2)).to_numpy()
        assert rel(col, Mo.w * Q.F2(Mo.x, Dz)[:, 1, 2]) < KTOL
        # the same function as an anonymous closure: host fallback, same values
        yh = M.map_rows(lambda xr, dr: F1(xr, dr), M.HPCMatrix(Mo.x[:5]), M.HPCMatrix(Dz[:5]))
        assert rel(yh.to_numpy(), Q.F1(Mo.x[:5], Dz[:5])) < KTOL
    outside = Dz.copy()
    outside[3, 3] = -1.0                                                    # s < 0: outside the cone -> F = +inf, reported not raised
    yo = M.map_rows(M.barrier_functions([([1, 2, 3], 1.5)], 4)[0], g.x, M.HPCMatrix(outside)).to_numpy()
    assert np.isinf(yo[3]) and np.isfinite(np.delete(yo, 3)).all()


def test_end_point_does_not_depend_on_summation_order(M):
    """The continuation ends at a fixed t, so z must not depend on mathematically neutral choices that
    change rounding and with it the kappa history: another split of the backward sweep, another elimination tree, separate
    instead of fused objective kernels.  fem2d L=6, p=1 (total variation: the ill-conditioned case), fresh process per variant
    (the knobs are read once); before the fixed end point such variants could land 1e-6 apart (profiles/r2_robust_sweep.txt)."""
    import subprocess
    import sys
    root = os.path.dirname(HERE)
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "robust_sweep.py"), "6", "1.0"], capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stderr[-500:]
    rows = [ln for ln in r.stdout.splitlines() if "|z - z_default|" in ln]
    assert len(rows) == 8, r.stdout
    for ln in rows:
        assert "t_final 1e+08" in ln, ln
        assert float(ln.rsplit("=", 1)[1]) < ZTOL, ln
    # two panels per launch (front_step2) against one, the two panels as a panel + an update launch, three single-panel tiles
    # per CU: the same factorisation bit for bit, hence the same solve
    for name in ("one_panel_steps", "panel_update_pairs", "dense_single_tiles"):
        one = [ln for ln in rows if ln.startswith(name)]
        assert len(one) == 1 and float(one[0].rsplit("=", 1)[1]) == 0.0, one


@pytest.mark.parametrize("kind,L,p", [("fem1d", 4, 2.0), ("fem2d", 3, 1.5)])
def test_stop_rules_visit_the_nominal_sequence(M, kind, L, p):
    """ADVICE r2 (mgb_amg_set_stop_rule): "upstream" = the literal `while t <= 1/tol: t <- kappa t`, "fixed" = end at the first
    t0 kappa^k beyond 1/tol.  Without a kappa reduction both give ts = t0 kappa^k exactly (SOL_main.ts is an observable of the
    reference, docs/src/api.md:97-101) and the same z, and the product follows the oracle under either rule."""
    tol = np.sqrt(np.finfo(np.float64).eps)
    want = [0.1]
    while want[-1] <= 1 / tol:
        want.append(want[-1] * 10.0)
    zs = {}
    for rule in ("fixed", "upstream"):
        sol = getattr(M, kind + "_mpi_solve")(L=L, p=p, stop_rule=rule)
        assert np.array_equal(sol.SOL_main["ts"], np.array(want)), rule
        zs[rule] = M.mpi_to_native(sol).z
        zo = getattr(O, kind + "_solve")(L=L, p=p, stop_rule=rule).z
        assert rel(zs[rule], zo) < ZTOL
    assert np.array_equal(zs["fixed"], zs["upstream"])
    with pytest.raises(ValueError):
        M.fem1d_mpi_solve(L=2, stop_rule="sometimes")


@pytest.mark.parametrize("kind,L,p", [("fem2d", 4, 1.5), ("fem2d", 5, 1.0), ("fem3d", 2, 1.0)])
def test_decrement_centering_matches_oracle_and_the_exact_rule(M, kind, L, p):
    """intermediate centres followed with the Newton-decrement rule (centering="decrement", oracle CENTERING),
    the last one resolved: same end point as the default rule and as the oracle run with the same rule."""
    z_exact = M.mpi_to_native(getattr(M, kind + "_mpi_solve")(L=L, p=p)).z
    sol = getattr(M, kind + "_mpi_solve")(L=L, p=p, centering="decrement")
    z = M.mpi_to_native(sol).z
    assert rel(z, z_exact) < ZTOL
    saved = O.CENTERING
    try:
        O.CENTERING = "decrement"
        so = getattr(O, kind + "_solve")(L=L, p=p)
    finally:
        O.CENTERING = saved
    assert rel(z, so.z) < ZTOL
    assert abs(int(sol.SOL_main["its"].sum()) - int(so.SOL_main["its"].sum())) <= max(3, 0.25 * so.SOL_main["its"].sum())
    with pytest.raises(ValueError):
        M.fem1d_mpi_solve(L=2, centering="never")


@pytest.mark.parametrize("kind,L", [("fem2d", 3), ("fem1d", 5), ("fem3d", 2)])
def test_x_dependent_exponent_matches_oracle(M, kind, L):
    """SURVEY.md section 8 f3 / VERDICT r2 item 9: p(x) -- per-node a = 2 / p(x_q) and mu(p(x_q)) read by the barrier kernels
    (mgb_amg_set_exponents).  Kernel level at every level (f0 / f1 / f2 against the oracle at 1e-12 / 1e-11), a constant p(x) bit for
    bit the scalar kernels, and the whole solve against the oracle (z at 1e-10).  p crosses 2, so mu takes all three values."""
    pfun = lambda x: 1.6 + 0.5 * x[0]                       # 1.1 .. 2.1 on [-1, 1]
    go = getattr(O, kind)(L) if kind != "fem3d" else O.fem3d(L)
    gm = getattr(M, kind + "_mpi")(L)
    dim = go.x.shape[1] if go.x.ndim > 1 else 1
    xo = go.x.reshape(go.x.shape[0], -1)
    pn = np.array([pfun(xi) for xi in xo])
    Mo = O.amg(go)
    K = len(Mo.D)
    idx = list(range(K - dim - 1, K))
    A = M.AMG(gm, p=1.0, cones=[(idx, pfun)])
    Ac = M.AMG(gm, p=1.0, cones=[(idx, np.full(xo.shape[0], 1.5))])
    As = M.AMG(gm, p=1.5)
    z0 = O.map_rows(lambda xi: O.DEFAULT_G[dim](xi), Mo.x).reshape(-1, order="F")
    c = O.map_rows(lambda xi: O.DEFAULT_F[dim](xi), Mo.x)
    for X in (A, Ac, As):
        X.set_c(c)
        X.set_z(z0)
    B = O.Barrier(O.convex_Euclidian_power(idx, pn))
    rng = np.random.default_rng(3)
    t = 2.5
    for l in range(L):
        Ro = Mo.R[l]
        Rg = sp.block_diag([gm.subspaces["dirichlet"][l].host, gm.subspaces["full"][l].host], format="csr")
        pi = _match_columns(Ro, Rg)
        N = Ro.shape[1]
        so = 2e-3 * rng.standard_normal(N)
        sg = np.zeros(N)
        sg[pi] = so
        y_o = B.f0(so, Mo.x, Mo.w, t * c, Ro, Mo.D, z0)
        assert abs(A.f0(l, sg, t) - y_o) <= KTOL * abs(y_o)
        assert rel(A.f1(l, sg, t)[pi], B.f1(so, Mo.x, Mo.w, t * c, Ro, Mo.D, z0)) < 1e-11
        H_o = B.f2(so, Mo.x, Mo.w, t * c, Ro, Mo.D, z0).toarray()
        H_g = A.f2(l, sg, t)[0].toarray()[np.ix_(pi, pi)]
        assert np.abs(H_g - H_o).max() <= 1e-11 * np.abs(H_o).max()
        assert Ac.f0(l, sg, t) == As.f0(l, sg, t) and np.array_equal(Ac.f1(l, sg, t), As.f1(l, sg, t))
        assert np.array_equal(Ac.f2(l, sg, t)[1], As.f2(l, sg, t)[1])
    sol = getattr(M, kind + "_mpi_solve")(L=L, p=pfun)
    so = O.amgb(go, p=pfun)
    assert rel(M.mpi_to_native(sol).z, so.z) < ZTOL


@pytest.mark.parametrize("kind,L,p", [("fem1d", 3, 1.0), ("fem2d", 2, 2.0), ("fem2d", 3, 1.0), ("fem2d", 3, 1.5), ("fem3d", 2, 1.0)])
def test_float32_solve_at_the_reference_float32_tolerance(M, kind, L, p):
    """SURVEY.md section 8 f3 / VERDICT r2 item 9: a Float32 SOLVE.  The reference's Float32 configurations (Metal backend,
    test/test_utils.jl:67-88) are held to 1e-4 (test_utils.jl:118-119) at tol = sqrt(eps(Float32)); here the float instantiation
    of the kernels drives the Newton loop (double device Cholesky on the float-assembled values) and z is compared with the
    double oracle at the same tol.  Measured: 1e-7 .. 6e-7."""
    tol32 = float(np.sqrt(np.finfo(np.float32).eps))
    sol = getattr(M, kind + "_mpi_solve")(L=L, p=p, T=np.float32)
    z = M.mpi_to_native(sol).z
    zo = getattr(O, kind + "_solve")(L=L, p=p, tol=tol32).z
    assert sol.SOL_main["T"] == "float32" and np.array_equal(z, z.astype(np.float32).astype(np.float64))
    assert sol.SOL_main["ts"][-1] == 1e4 and rel(z, zo) < 1e-4
    print("%s L=%d p=%g float32: newton %d, rel l2 vs the double oracle at tol32 %.2e" % (kind, L, p, int(sol.SOL_main["its"].sum()), rel(z, zo)))


@pytest.mark.parametrize("kind,L,p,psi", [("fem2d", 3, 1.5, 0.1), ("fem2d", 4, 2.0, 0.2), ("fem1d", 5, 1.5, 0.1)])
def test_piecewise_set_matches_oracle(M, kind, L, p, psi):
    """upstream `convex_piecewise` -- a convex set that varies in space ([UPSTREAM-UNVERIFIED] semantics: at x the
    intersection of the pieces select(x) keeps).  Here the p-Laplace cone everywhere and the obstacle u > psi only on the half
    x_1 > 0 (mgb_amg_set_term_mask; oracle ConvexPiecewise): kernel level at 1e-12 / 1e-11, the solve at 1e-10; the obstacle is
    active where it applies and violated where it does not."""
    dim = 1 if kind == "fem1d" else 2
    gm = getattr(M, kind + "_mpi")(L)
    go = getattr(O, kind)(L)
    cone = (list(range(1, dim + 2)), p)
    lin = ("linear", [0], [1.0], -psi)
    select = lambda x: (True, x[0] > 0.0)
    A = M.AMG(gm, p=p, cones=[cone, lin], select=select)
    Mo = O.amg(go)
    z0 = O.map_rows(lambda xi: OBST_G[dim](xi), Mo.x).reshape(-1, order="F")
    c = O.map_rows(lambda xi: OBST_F[dim](xi), Mo.x)
    A.set_c(c)
    A.set_z(z0)
    Bo = O.Barrier(O.convex_piecewise([O.convex_Euclidian_power(list(range(1, dim + 2)), p), O.LinearBarrier([0], [1.0], -psi)],
                                      select, Mo.x.reshape(Mo.x.shape[0], -1)))
    l = L - 1
    Ro = Mo.R[l]
    Rg = sp.block_diag([gm.subspaces["dirichlet"][l].host, gm.subspaces["full"][l].host], format="csr")
    pi = _match_columns(Ro, Rg)
    N = Ro.shape[1]
    rng = np.random.default_rng(3)
    so = 2e-3 * rng.standard_normal(N)
    sg = np.zeros(N)
    sg[pi] = so
    t = 2.5
    y_o = Bo.f0(so, Mo.x, Mo.w, t * c, Ro, Mo.D, z0)
    assert np.isfinite(y_o) and abs(A.f0(l, sg, t) - y_o) <= KTOL * abs(y_o)
    assert rel(A.f1(l, sg, t)[pi], Bo.f1(so, Mo.x, Mo.w, t * c, Ro, Mo.D, z0)) < 1e-11
    H_o = Bo.f2(so, Mo.x, Mo.w, t * c, Ro, Mo.D, z0).toarray()
    H_g = A.f2(l, sg, t)[0].toarray()[np.ix_(pi, pi)]
    assert np.abs(H_g - H_o).max() <= 1e-11 * np.abs(H_o).max()
    sol = M.amgb(gm, p=p, f=OBST_F[dim], g=OBST_G[dim], cones=[cone, lin], select=select)
    ref = O.amgb(go, p=p, f=OBST_F[dim], g=OBST_G[dim], extra=[O.LinearBarrier([0], [1.0], -psi)], select=select)
    z = M.mpi_to_native(sol).z
    assert rel(z, ref.z) < ZTOL
    x1 = Mo.x.reshape(Mo.x.shape[0], -1)[:, 0]
    on, off = z[x1 > 0, 0], z[x1 < 0, 0]
    assert on.min() > psi and on.min() - psi < 1e-3 and off.min() < psi       # active where selected, ignored elsewhere
    with pytest.raises(M._lib.MGBError):
        M.AMG(gm, p=p, cones=[cone, lin], select=lambda x: (False, False))
