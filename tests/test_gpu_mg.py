"""GPU tests of the multigrid pieces (`-m gpu`, through the C ABI): matrix-free Hessian product,
Chebyshev-Jacobi smoother, prolongation / restriction, V-cycle-preconditioned CG as the Newton linear solver.

The reference has no smoother (its levels are solved directly, test/test_instrumented_solve.jl:25-28,99), so these pieces are
held to (a) the operator the reference's Hessian recipe defines -- the oracle's f2 matrix, test/test_map_rows_compare.jl:102-179
-- at 1e-12, (b) a plain numpy restatement of the same smoother recurrence written here, and (c) end-to-end parity of
solver="pcg" solves with the same oracle vectors and goldens the direct solver is held to (z at 1e-10)."""
import os

import numpy as np
import pytest
import scipy.sparse as sp
import scipy.sparse.linalg as spla

import mgb_oracle as O
from test_gpu_parity import _match_columns, _problem, rel, LARGE_CASES, LARGE_CENTRE_TOL, LARGE_END_POINT_TOL, ZTOL

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def M(gpu_required):
    import mgb_amd
    return mgb_amd


def _level_maps(A, Mo, l):
    sub = A.geometry.subspaces
    Ro = Mo.R[l]
    Rg = sp.block_diag([sub["dirichlet"][l].host, sub["full"][l].host], format="csr")
    return Ro, Rg, _match_columns(Ro, Rg)


@pytest.mark.parametrize("kind,L,p", [("fem1d", 4, 1.0), ("fem1d", 5, 1.5), ("fem2d", 2, 1.0), ("fem2d", 3, 1.5), ("fem2d", 4, 1.0),
                                      ("fem2d", 3, 3.0), ("fem3d", 2, 1.5), ("fem3d", 2, 1.0)])
def test_hessian_apply_matches_oracle_all_levels(M, kind, L, p):
    """H v = B' (Y o (B v)) on the device (element-local, matrix-free) and through the assembled matrix, against the oracle's
    f2 matrix times v (the reference's recipe, test/test_map_rows_compare.jl:102-123,165-170) at every level."""
    A, Mo, B, z0, c, go = _problem(M, kind, L, p)
    rng = np.random.default_rng(5)
    for l in range(L):
        Ro, Rg, pi = _level_maps(A, Mo, l)
        N = Ro.shape[1]
        so = 2e-3 * rng.standard_normal(N)
        vo = rng.standard_normal(N)
        sg, vg = np.zeros(N), np.zeros(N)
        sg[pi], vg[pi] = so, vo
        H_o = B.f2(so, Mo.x, Mo.w, 3.7 * c, Ro, Mo.D, z0)
        want = H_o @ vo
        got = A.hessian_apply(l, sg, vg, matrix_free=True)
        assert rel(got[pi], want) < 1e-12
        got_a = A.hessian_apply(l, sg, vg, matrix_free=False)
        assert rel(got_a[pi], want) < 1e-12
        # symmetry of the operator as applied: <u, H v> = <v, H u>
        ug = rng.standard_normal(N)
        assert abs(ug @ got - vg @ A.hessian_apply(l, sg, ug)) <= 1e-11 * abs(ug @ got)


def test_hessian_apply_is_reproducible(M):
    A, Mo, B, z0, c, go = _problem(M, "fem2d", 4, 1.0)
    l = 3
    N = A.level_size(l)[0]
    rng = np.random.default_rng(1)
    s, v = 1e-3 * rng.standard_normal(N), rng.standard_normal(N)
    a = A.hessian_apply(l, s, v)
    assert all(np.array_equal(a, A.hessian_apply(l, s, v)) for _ in range(3))      # gather form, fixed order: bit for bit


@pytest.mark.parametrize("kind,L", [("fem1d", 5), ("fem2d", 4), ("fem3d", 2)])
def test_prolongation_nests_the_levels(M, kind, L):
    """R_l = R_{l+1} P_l (the AMG levels of test/test_d0_construction.jl:82 are nested), and the device transfer kernels apply
    P and P'."""
    A, Mo, B, z0, c, go = _problem(M, kind, L, 1.0)
    sub = A.geometry.subspaces
    rng = np.random.default_rng(2)
    for l in range(L - 1):
        Rc = sp.block_diag([sub["dirichlet"][l].host, sub["full"][l].host], format="csr")
        Rf = sp.block_diag([sub["dirichlet"][l + 1].host, sub["full"][l + 1].host], format="csr")
        P = A.prolongation(l)
        assert P.shape == (Rf.shape[1], Rc.shape[1])
        assert abs(Rf @ P - Rc).max() < 1e-12
        xc = rng.standard_normal(Rc.shape[1])
        rf = rng.standard_normal(Rf.shape[1])
        assert rel(A.prolong(l, xc), P @ xc) < 1e-14
        assert rel(A.restrict(l, rf), P.T @ rf) < 1e-13


def _cheb_numpy(H, b, x, lam, degree, sweeps, lo_frac=0.12, hi_frac=1.2):
    """Chebyshev iteration for the Jacobi-preconditioned system (Saad, Iterative Methods, Alg. 12.1) on [lo, hi] * lam:
    the recurrence csrc/mg.hip runs, `degree` applications of H per sweep."""
    dinv = 1.0 / H.diagonal()
    hi, lo = hi_frac * lam, lo_frac * lam
    theta, delta = 0.5 * (hi + lo), 0.5 * (hi - lo)
    sigma = theta / delta
    for _ in range(sweeps):
        r = b - H @ x
        rho = 1.0 / sigma
        d = dinv * r / theta
        for k in range(1, degree):
            x = x + d
            r = r - H @ d
            rn = 1.0 / (2.0 * sigma - rho)
            d = rn * rho * d + (2.0 * rn / delta) * (dinv * r)
            rho = rn
        x = x + d
    return x


@pytest.mark.parametrize("kind,L,p,l", [("fem2d", 4, 1.0, 3), ("fem2d", 4, 1.5, 2), ("fem1d", 5, 2.0, 4), ("fem3d", 2, 1.5, 1)])
@pytest.mark.parametrize("matrix_free", [True, False])
def test_smoother_matches_numpy_restatement(M, kind, L, p, l, matrix_free):
    A, Mo, B, z0, c, go = _problem(M, kind, L, p)
    Ro, Rg, pi = _level_maps(A, Mo, l)
    N = Ro.shape[1]
    rng = np.random.default_rng(9)
    so = 1e-3 * rng.standard_normal(N)
    sg = np.zeros(N)
    sg[pi] = so
    H = sp.csr_matrix(B.f2(so, Mo.x, Mo.w, 2.0 * c, Ro, Mo.D, z0))
    d = H.diagonal()
    lam = float(spla.eigsh(sp.diags(d ** -0.5) @ H @ sp.diags(d ** -0.5), k=1, which="LA", return_eigenvectors=False)[0])
    bo = rng.standard_normal(N)
    bg = np.zeros(N)
    bg[pi] = bo
    for degree, sweeps in ((1, 2), (2, 1), (3, 2), (4, 1)):
        want = _cheb_numpy(H, bo, np.zeros(N), lam, degree, sweeps)
        got, used = A.smooth(l, sg, bg, degree=degree, sweeps=sweeps, lmax=lam, matrix_free=matrix_free)
        assert used == lam
        assert rel(got[pi], want) < 1e-11
    # the device's own estimate of lambda_max(Dinv H): power steps in the D inner product never overshoot
    x0 = np.zeros(N)
    _, est = A.smooth(l, sg, bg, x0, degree=2, sweeps=1, lmax=0.0, matrix_free=matrix_free)
    assert 0.9 * lam <= est <= lam * (1 + 1e-10)
    # and the smoother smooths: the error of a rough vector shrinks in the energy norm
    xs = spla.spsolve(H.tocsc(), bo)
    e0 = -xs
    xg, _ = A.smooth(l, sg, bg, degree=3, sweeps=2, lmax=0.0, matrix_free=matrix_free)
    e1 = xg[pi] - xs
    assert e1 @ (H @ e1) < e0 @ (H @ e0)


@pytest.mark.parametrize("kind,L,p", [("fem2d", 4, 1.0), ("fem2d", 5, 1.5), ("fem1d", 7, 1.0), ("fem3d", 3, 1.5), ("fem2d", 2, 2.0)])
def test_pcg_linear_solve_matches_direct(M, kind, L, p):
    """V-cycle-preconditioned CG on one Newton system against the oracle's direct solve (MultiGridBarrier.solve = A \\ b)."""
    A, Mo, B, z0, c, go = _problem(M, kind, L, p)
    l = L - 1
    Ro, Rg, pi = _level_maps(A, Mo, l)
    N = Ro.shape[1]
    rng = np.random.default_rng(4)
    so = 1e-3 * rng.standard_normal(N)
    sg = np.zeros(N)
    sg[pi] = so
    t = 10.0
    H = sp.csc_matrix(B.f2(so, Mo.x, Mo.w, t * c, Ro, Mo.D, z0))
    g_o = B.f1(so, Mo.x, Mo.w, t * c, Ro, Mo.D, z0)
    gg = np.zeros(N)
    gg[pi] = g_o
    want = spla.spsolve(H, g_o)
    A.set_pcg(rtol=1e-11, maxit=300)
    x, it, rr, ok = A.pcg_solve_linear(l, sg, gg)
    print("%s L=%d p=%g: N=%d coarsest level %d, CG iterations %d, M-norm residual %.2e" % (kind, L, p, N, A.mg_coarsest(l), it, rr))
    assert ok and it <= 300
    assert rel(x[pi], want) < 1e-8
    # the assembled-top variant solves the same system
    A.set_pcg(assembled_top=True)
    xa, ita, _, oka = A.pcg_solve_linear(l, sg, gg)
    assert oka and rel(xa[pi], want) < 1e-8
    A.set_pcg(assembled_top=False)


@pytest.mark.parametrize("kind,L,p", [("fem1d", 4, 2.0), ("fem2d", 3, 1.0), ("fem2d", 3, 2.0), ("fem3d", 2, 1.0)])
def test_pcg_solve_matches_golden(M, kind, L, p):
    sol = getattr(M, kind + "_mpi_solve")(L=L, p=p, solver="pcg")
    z = M.mpi_to_native(sol).z
    gold = np.load(os.path.join(HERE, "golden", "%s_L%d_p%s.npz" % (kind, L, str(p).replace(".", "_"))))
    pc = sol.SOL_main["pcg"]
    print("%s L=%d p=%g pcg: newton %d, CG iterations %d (%.1f per system), fallbacks %d, rel l2 %.2e"
          % (kind, L, p, int(sol.SOL_main["its"].sum()), pc["iterations"], pc["iterations"] / max(pc["solves"], 1), pc["fallbacks"],
             rel(z, gold["z"])))
    assert pc["solves"] > 0
    assert rel(z, gold["z"]) < ZTOL


@pytest.mark.parametrize("kind,L,p", [("fem2d", 5, 1.5), ("fem2d", 5, 1.0)])
def test_pcg_solve_matches_live_oracle(M, kind, L, p):
    zo = getattr(O, kind + "_solve")(L=L, p=p).z
    sol = getattr(M, kind + "_mpi_solve")(L=L, p=p, solver="pcg")
    z = M.mpi_to_native(sol).z
    pc = sol.SOL_main["pcg"]
    print("%s L=%d p=%g pcg: newton %d, CG iterations %d (%.1f per system), fallbacks %d, rel l2 %.2e"
          % (kind, L, p, int(sol.SOL_main["its"].sum()), pc["iterations"], pc["iterations"] / max(pc["solves"], 1), pc["fallbacks"],
             rel(z, zo)))
    assert rel(z, zo) < ZTOL


@pytest.mark.parametrize("kind,L,p", LARGE_CASES)
def test_pcg_headline_sizes_match_oracle_goldens(M, kind, L, p):
    """solver="pcg" at the BASELINE sizes (fem2d L=7 p = 1 / 1.5, fem3d L=4) against the same committed oracle vectors as the
    direct solver (test_headline_sizes_match_oracle_goldens): the exact centre at 1e-11, u at 1e-10."""
    gold = np.load(os.path.join(HERE, "golden", "large_%s_L%d_p%s.npz" % (kind, L, str(p).replace(".", "_"))))
    sol = getattr(M, kind + "_mpi_solve")(L=L, p=p, solver="pcg")
    z = M.mpi_to_native(sol).z
    pc = sol.SOL_main["pcg"]
    err, err_centre = rel(z, gold["z"]), rel(z, gold["z_centre"])
    print("%s L=%d p=%g pcg: %.3f s, newton %d, CG iterations %d (%.1f per system), fallbacks %d; to the oracle end point %.3e, "
          "to the exact centre %.3e" % (kind, L, p, sol.SOL_main["t_elapsed"], int(sol.SOL_main["its"].sum()), pc["iterations"],
                                        pc["iterations"] / max(pc["solves"], 1), pc["fallbacks"], err, err_centre))
    assert err_centre < LARGE_CENTRE_TOL
    assert err < LARGE_END_POINT_TOL[kind]
    assert rel(z[:, 0], gold["z"][:, 0]) < ZTOL


def test_pcg_rejects_sharded_contexts_and_bad_parameters(M):
    g = M.fem2d_mpi(2)
    A = M.AMG(g, p=1.0)
    with pytest.raises(M._lib.MGBError):
        A.set_pcg(degree=9)
    with pytest.raises(ValueError):
        A.set_solver("jacobi")
