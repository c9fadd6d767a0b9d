"""GPU parity of the two-cone barrier kernels and of parabolic_solve (BASELINE config 5; reference test
test/test_parabolic.jl:41-104: ParabolicSOL shape, ts, per-snapshot comparison at 1e-10)."""
import numpy as np
import pytest
import scipy.sparse as sp

import mgb_oracle as O

pytestmark = pytest.mark.gpu
KTOL = 1e-12


@pytest.fixture(scope="module")
def M(gpu_required):
    import mgb_amd
    return mgb_amd


def rel(a, b):
    a, b = np.asarray(a, dtype=float), np.asarray(b, dtype=float)
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


@pytest.mark.parametrize("kind,L,p", [("fem1d", 3, 2.0), ("fem2d", 2, 1.0), ("fem2d", 3, 1.5)])
def test_two_cone_kernels_match_oracle(M, kind, L, p):
    go = getattr(O, kind)(L)
    gm = getattr(M, kind + "_mpi")(L)
    dim = go.discretization["dim"]
    state, D, K, cones, ops = O.parabolic_problem(go, p)
    A = M.AMG(gm, state, D, p, cones=cones)
    assert A.nY == 3 + (dim + 1) * (dim + 2) // 2
    Mo = O.amg(go, state, D)
    B = O.Barrier(O.ConeIntersection([O.convex_Euclidian_power(i, pp) for i, pp in cones]))
    z0 = O.parabolic_initial(go, p, O.DEFAULT_G[dim])
    n = go.x.shape[0]
    rng = np.random.default_rng(5)
    c = O.parabolic_cost(n, K, p, 0.25, np.full(n, 0.5), z0[:n] + 0.1 * rng.standard_normal(n))
    A.set_c(c)
    A.set_z(z0)
    t = 2.3
    for l in range(L):
        Ro = Mo.R[l]
        subs = gm.subspaces
        Rg = sp.block_diag([subs["dirichlet"][l].host, subs["full"][l].host, subs["full"][l].host], format="csr")
        f = np.sin(np.arange(Ro.shape[0]) * 0.7 + 1.0)
        po, pg = np.argsort(Ro.T @ f), np.argsort(Rg.T @ f)
        pi = np.empty(len(po), dtype=int)
        pi[po] = pg
        assert abs(Ro - Rg[:, pi]).max() < 1e-12
        N = Ro.shape[1]
        so = 1e-3 * rng.standard_normal(N)
        sg = np.zeros(N)
        sg[pi] = so
        yo = B.f0(so, Mo.x, Mo.w, t * c, Ro, Mo.D, z0)
        assert np.isfinite(yo)
        assert abs(A.f0(l, sg, t) - yo) <= KTOL * abs(yo)
        assert rel(A.f1(l, sg, t)[pi], B.f1(so, Mo.x, Mo.w, t * c, Ro, Mo.D, z0)) < 1e-11
        Ho = B.f2(so, Mo.x, Mo.w, t * c, Ro, Mo.D, z0).toarray()
        Hg, lower = A.f2(l, sg, t)
        assert np.abs(Hg.toarray()[np.ix_(pi, pi)] - Ho).max() <= 1e-11 * np.abs(Ho).max()


@pytest.mark.parametrize("kind,L,p,h", [("fem1d", 2, 2.0, 0.5), ("fem1d", 4, 1.0, 0.25), ("fem2d", 2, 1.5, 0.5),
                                        ("fem2d", 3, 1.0, 0.5)])
def test_parabolic_matches_oracle(M, kind, L, p, h):
    """test/test_parabolic.jl:48 runs fem1d L=2, h=0.5, t1=1, p=2; each snapshot must match to 1e-10."""
    g = getattr(M, kind + "_mpi")(L)
    sol = M.parabolic_solve(g, h=h, t1=1.0, p=p, verbose=False)
    assert isinstance(sol, M.ParabolicSOL) and sol.geometry is g            # :50-51
    assert len(sol.ts) >= 2 and len(sol.u) == len(sol.ts)                    # :52-53
    assert isinstance(sol.u[0], M.HPCMatrix) and isinstance(sol.u[-1], M.HPCMatrix)   # :63-64
    nat = M.mpi_to_native(sol)
    assert isinstance(nat, M.ParabolicSOL) and isinstance(nat.geometry.x, np.ndarray)
    assert isinstance(nat.u[0], np.ndarray) and np.array_equal(nat.ts, sol.ts)        # :73-78
    ref = O.parabolic_solve(getattr(O, kind)(L), h=h, t1=1.0, p=p)
    assert np.array_equal(nat.ts, ref.ts)
    for uk, rk in zip(nat.u, ref.u):
        assert uk.shape == rk.shape
        # u itself at the reference's 1e-10 (:93-104); the slack columns sit within 1/t = 1e-8 of their cones
        # along flat directions of the objective and are only determined to O(cond * eps)
        assert rel(uk[:, 0], rk[:, 0]) < 1e-10
        assert rel(uk, rk) < 1e-8


def test_parabolic_properties_2d_L6(M):
    """BASELINE config 5 (2-D parabolic, L=6): energy-type properties, no oracle run at this size."""
    g = M.fem2d_mpi(6)
    sol = M.mpi_to_native(M.parabolic_solve(g, h=0.25, t1=1.0, p=2.0))
    ops = sol.geometry.operators
    w = sol.geometry.w
    assert len(sol.u) == 5
    for z in sol.u[1:]:
        grad2 = (ops["dx"] @ z[:, 0]) ** 2 + (ops["dy"] @ z[:, 0]) ** 2
        assert np.all(z[:, 1] > z[:, 0] ** 2) and np.all(z[:, 2] > grad2)     # strictly inside both cones
        # at t = 1e8 the slacks are tight: s1 everywhere (u is continuous), s2 where the continuous slack meets
        # the largest of the adjacent elements' discontinuous |grad u|^2
        assert (z[:, 1] - z[:, 0] ** 2).max() < 1e-5 and (z[:, 2] - grad2).min() < 1e-6
    # implicit Euler is a minimising movement: E(z) = int (1/p) s2 + f1 u cannot increase from one minimiser
    # to the next (E(z_{k+1}) + |u_{k+1}-u_k|^2/(2h) <= E(z_k))
    E = [float(np.dot(w, z[:, 2] / 2.0 + 0.5 * z[:, 0])) for z in sol.u[1:]]
    assert all(b <= a + 1e-6 for a, b in zip(E, E[1:]))


def test_parabolic_2d_L6_matches_oracle_golden(M):
    """BASELINE config 5 at full size (2-D parabolic, L=6, h=0.1, t1=1, p=1 as in docs/src/guide.md:367): snapshots 1, 5
    and 10 of the ten implicit-Euler steps against the CPU oracle's (tests/golden/make_golden_large.py)."""
    import os
    gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "large_parabolic_L6_p1_0.npz"))
    sol = M.mpi_to_native(M.parabolic_solve(M.fem2d_mpi(6), h=0.1, t1=1.0, p=1.0))
    assert np.allclose(sol.ts, gold["ts"], rtol=0, atol=1e-15) and len(sol.u) == 11
    for k in gold["keep"]:
        uk, rk = sol.u[int(k)], gold["u_%d" % int(k)]
        print("parabolic L=6 snapshot %d: u rel l2 %.3e, all columns %.3e" % (k, rel(uk[:, 0], rk[:, 0]), rel(uk, rk)))
        assert rel(uk[:, 0], rk[:, 0]) < 1e-10          # u at the reference's bar (test/test_parabolic.jl:93-104); measured <= 1e-12
        assert rel(uk, rk) < 1e-9                       # slack columns: flat directions, O(cond * eps); measured <= 2.5e-11


@pytest.mark.parametrize("kind,L,p", [("fem1d", 3, 1.0), ("fem2d", 3, 1.0), ("fem2d", 2, 2.0)])
def test_feasibility_phase(M, kind, L, p):
    """SOL_feasibility (src:428-455): an infeasible start (slack too small) is repaired by the feasibility
    phase and the main phase lands on the same z as the oracle; a feasible start keeps SOL_feasibility None."""
    gbad = {1: lambda x: np.array([x[0], 0.2]), 2: lambda x: np.array([x[0] ** 2 + x[1] ** 2, 0.5])}
    dim = 1 if kind == "fem1d" else 2
    sol = getattr(M, kind + "_mpi_solve")(L=L, p=p, g=gbad[dim])
    assert sol.SOL_feasibility is not None and sol.SOL_feasibility["shift"] > 0
    ref = O.amgb(getattr(O, kind)(L), p=p, g=gbad[dim])
    assert ref.SOL_feasibility is not None
    assert abs(sol.SOL_feasibility["shift"] - ref.SOL_feasibility["shift"]) <= 1e-12 * ref.SOL_feasibility["shift"]
    assert rel(M.mpi_to_native(sol).z, ref.z) < 1e-10
    assert getattr(M, kind + "_mpi_solve")(L=L, p=p).SOL_feasibility is None        # src:428-430
