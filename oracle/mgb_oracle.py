"""CPU oracle (TEST INFRASTRUCTURE ONLY) for the multigrid-barrier Newton path.

This file is a numpy/scipy restatement of the algorithm behind
``fem{1,2}d_mpi_solve`` / ``amgb`` of sloisel/MultiGridBarrierMPI.jl.  It is the
*checker*: only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` may import it.  The product path (``multigridbarriermpi.jl_amd``)
never does.

PARITY STATUS: "parity unpinned" at solve level.
  The reference (/root/reference) is pure Julia glue; the arithmetic lives in the
  un-vendored dependencies MultiGridBarrier.jl (compat 0.11, Project.toml:33) and
  HPCSparseArrays.jl (compat 0.1, Project.toml:29).  Julia is not installed and the
  reference holds no stored solve outputs.  What *is* pinned here
  (tests/test_oracle_kats.py) are the reference's own known-answer tests:
    * map_rows exact values            test/test_helpers.jl:129-167, test/test_map_rows.jl:27-101
    * amgb_all_isfinite / amgb_diag / amgb_zeros / amgb_blockdiag
                                        test/test_helpers.jl:53-121, test/test_diag.jl:28-46
    * the Hessian recipe identity       test/test_matrix_addition.jl:39-95,
                                        test/test_d0_construction.jl:92-185,
                                        test/test_map_rows_compare.jl:102-179
    * structure: fem1d L=3 -> 16 rows, R 16x7 (test/test_nonsquare.jl:28);
      fem1d L=2 -> 8 rows (test/test_partition_debug.jl:34);
      fem2d n = 14*4^(L-1) (docs/src/guide.md:246-253).
  Everything else (node ordering, quadrature, barrier constants, Newton/line-search/
  t-update rules) restates the published method (S. Loisel, multigrid barrier /
  p-Laplace papers) and is validated by PDE-level known answers, not reference numbers.

Conventions (shared with the HIP product):
  * broken (element-wise discontinuous) nodal storage: n rows = n_elements * nodes_per_element
  * z is an (n, S) matrix of state variables (u, s); its vectorisation is column-major
    ``[u; s]`` exactly as Julia's ``vec``/``hcat`` layout implies (test_d0_construction.jl:92-100).
  * ``subspaces[key][l]`` has n (finest) rows for every level l
    (test/test_hessian.jl:85-93 multiplies ``subspaces[:dirichlet][1]`` with the fine ``dx``).
"""
from __future__ import annotations

import math
import time
from dataclasses import dataclass, field
from typing import Callable, Dict, List, Optional, Sequence

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

# ----------------------------------------------------------------------------
# Hook semantics (reference src/MultiGridBarrierMPI.jl:62-192)
# ----------------------------------------------------------------------------


def map_rows(f: Callable, *arrays):
    """Row-wise map; follows src:161-163 + tools/profile_map_rows_steps.jl:57-146.

    Each argument contributes its i-th row (vector -> length-1 row, matrix -> row).
    Scalar results stack into a vector, row-vector results into a matrix.
    """
    A = [np.asarray(a, dtype=np.float64) for a in arrays]
    n = A[0].shape[0]
    rows = []
    for i in range(n):
        args = [a[i:i + 1] if a.ndim == 1 else a[i, :] for a in A]
        rows.append(np.asarray(f(*args), dtype=np.float64))
    if rows and rows[0].ndim == 0:
        return np.array([float(r) for r in rows])
    return np.vstack([r.reshape(1, -1) for r in rows]) if rows else np.zeros((0,))


def amgb_all_isfinite(z) -> bool:
    """src:121-133 (local all(isfinite) AND-reduced over ranks)."""
    return bool(np.all(np.isfinite(np.asarray(z))))


def amgb_diag(z, m=None, n=None):
    """src:137-147: spdiagm(m, n, 0 => z)."""
    z = np.asarray(z, dtype=np.float64)
    m = len(z) if m is None else m
    n = len(z) if n is None else n
    return sp.diags([z], [0], shape=(m, n), format="csr")


def amgb_zeros(m, n=None):
    """src:66-75,116: sparse/dense zeros."""
    return np.zeros(m) if n is None else sp.csr_matrix((m, n))


def amgb_blockdiag(*mats):
    """src:150."""
    return sp.block_diag(mats, format="csr")


# ----------------------------------------------------------------------------
# Geometry  (MultiGridBarrier `Geometry` fields: src:318-330, docs/src/api.md:79-87)
# ----------------------------------------------------------------------------


@dataclass
class Geometry:
    discretization: dict
    x: np.ndarray            # (n, d) broken node coordinates
    w: np.ndarray            # (n,)   quadrature weights
    subspaces: Dict[str, List[sp.csr_matrix]]   # key -> [n x m_l] for l = 1..L
    operators: Dict[str, sp.csr_matrix]         # 'id','dx','dy'  (n x n)
    refine: List[sp.csr_matrix]                 # refine[l]: n_l -> n_{l+1}; refine[L-1] = I
    coarsen: List[sp.csr_matrix]                # coarsen[l] @ refine[l] = I


def fem1d(L: int = 4) -> Geometry:
    """1-D broken P1 elements on [-1,1]; 2^l elements at level l (src:547: "2^L elements").

    fem1d L=3 -> 16 rows and a 16x7 Dirichlet subspace (test/test_nonsquare.jl:28).
    """
    def level_nodes(l):
        ne = 2 ** l
        edges = np.linspace(-1.0, 1.0, ne + 1)
        return np.stack([edges[:-1], edges[1:]], axis=1).reshape(-1)  # (2*ne,)

    xs = [level_nodes(l) for l in range(1, L + 1)]
    n = xs[-1].size
    ne = 2 ** L
    h = 2.0 / ne
    w = np.full(n, h / 2)
    blk = np.array([[-1.0, 1.0], [-1.0, 1.0]]) / h
    dx = sp.block_diag([blk] * ne, format="csr")
    ident = sp.identity(n, format="csr")
    # refine[l]: level l+1 (1-based l) -> l+2 ; each element -> 2 children
    child = np.array([[1.0, 0.0], [0.5, 0.5], [0.5, 0.5], [0.0, 1.0]])
    refine, coarsen = [], []
    for l in range(1, L):
        nel = 2 ** l
        refine.append(sp.block_diag([child] * nel, format="csr"))
        inj = np.zeros((2, 4))
        inj[0, 0] = 1.0
        inj[1, 3] = 1.0
        coarsen.append(sp.block_diag([inj] * nel, format="csr"))
    refine.append(sp.identity(n, format="csr"))
    coarsen.append(sp.identity(n, format="csr"))
    full, dirichlet = [], []
    for l in range(1, L + 1):
        nel = 2 ** l
        # continuous dofs: nel+1 vertices
        rows = np.arange(2 * nel)
        cols = (rows + 1) // 2
        S = sp.csr_matrix((np.ones(2 * nel), (rows, cols)), shape=(2 * nel, nel + 1))
        P = S
        for k in range(l - 1, L - 1):
            P = refine[k] @ P
        P = sp.csr_matrix(P)
        full.append(P)
        dirichlet.append(sp.csr_matrix(P[:, 1:-1]))
    return Geometry(dict(kind="fem1d", L=L, dim=1, block=2), xs[-1].reshape(-1, 1), w,
                    dict(full=full, dirichlet=dirichlet), dict(id=ident, dx=dx), refine, coarsen)


# reference-triangle nodes (barycentric-free: xi, eta) in the order v1,v2,v3,m12,m23,m31,c
_TRI_NODES = np.array([[0, 0], [1, 0], [0, 1], [.5, 0], [.5, .5], [0, .5], [1 / 3, 1 / 3]])
_TRI_W = np.array([1 / 20] * 3 + [2 / 15] * 3 + [9 / 20])  # x area; exact for cubics


def _tri_basis(pts):
    """Values and (xi,eta)-derivatives of the 7 monomials spanning P2 + cubic bubble."""
    xi, et = pts[:, 0], pts[:, 1]
    lam = 1 - xi - et
    V = np.stack([np.ones_like(xi), xi, et, xi * xi, xi * et, et * et, xi * et * lam], axis=1)
    Vx = np.stack([0 * xi, 1 + 0 * xi, 0 * xi, 2 * xi, et, 0 * xi, et * lam - xi * et], axis=1)
    Vy = np.stack([0 * xi, 0 * xi, 1 + 0 * xi, 0 * xi, xi, 2 * et, xi * lam - xi * et], axis=1)
    return V, Vx, Vy


def _tri_ref_mats():
    V, Vx, Vy = _tri_basis(_TRI_NODES)
    C = np.linalg.inv(V)
    return C, Vx @ C, Vy @ C   # nodal coefficient map; d/dxi, d/deta at the nodes


def fem2d(L: int = 2, K: Optional[np.ndarray] = None) -> Geometry:
    """2-D broken P2+bubble triangles (7 nodes/element), red refinement.

    n = 14 * 4^(L-1) (docs/src/guide.md:246-253).  ``K`` = (3m x 2) vertex list of the
    coarse triangles (docs/src/guide.md:317); default = [-1,1]^2 split into 2 triangles.
    """
    if K is None:
        K = np.array([[-1, -1], [1, -1], [-1, 1], [1, -1], [1, 1], [-1, 1]], dtype=np.float64)
    K = np.asarray(K, dtype=np.float64)
    tris = [K.reshape(-1, 3, 2)]
    for _ in range(1, L):
        T = tris[-1]
        v1, v2, v3 = T[:, 0], T[:, 1], T[:, 2]
        m12, m23, m31 = (v1 + v2) / 2, (v2 + v3) / 2, (v3 + v1) / 2
        ch = np.stack([np.stack([v1, m12, m31], 1), np.stack([m12, v2, m23], 1),
                       np.stack([m31, m23, v3], 1), np.stack([m23, m31, m12], 1)], axis=1)
        tris.append(ch.reshape(-1, 3, 2))
    C, Dxi, Det = _tri_ref_mats()

    def nodes_of(T):
        v1, v2, v3 = T[:, 0], T[:, 1], T[:, 2]
        P = np.stack([v1, v2, v3, (v1 + v2) / 2, (v2 + v3) / 2, (v3 + v1) / 2, (v1 + v2 + v3) / 3], axis=1)
        return P.reshape(-1, 2)

    T = tris[-1]
    ne = T.shape[0]
    x = nodes_of(T)
    n = x.shape[0]
    e1, e2 = T[:, 1] - T[:, 0], T[:, 2] - T[:, 0]
    det = e1[:, 0] * e2[:, 1] - e1[:, 1] * e2[:, 0]
    area = np.abs(det) / 2
    w = (area[:, None] * _TRI_W[None, :]).reshape(-1)
    # x = v1 + e1*xi + e2*eta  ->  [dxi/dx dxi/dy; deta/dx deta/dy] = inv([e1 e2])
    xix, xiy = e2[:, 1] / det, -e2[:, 0] / det
    etx, ety = -e1[:, 1] / det, e1[:, 0] / det
    bx = xix[:, None, None] * Dxi[None] + etx[:, None, None] * Det[None]
    by = xiy[:, None, None] * Dxi[None] + ety[:, None, None] * Det[None]
    dx = sp.block_diag(list(bx), format="csr") if ne < 4096 else _bdiag(bx)
    dy = sp.block_diag(list(by), format="csr") if ne < 4096 else _bdiag(by)
    ident = sp.identity(n, format="csr")
    # refine / coarsen between consecutive levels
    child_ref = [np.array(c, dtype=np.float64) for c in (
        [[0, 0], [.5, 0], [0, .5]], [[.5, 0], [1, 0], [.5, .5]],
        [[0, .5], [.5, .5], [0, 1]], [[.5, .5], [0, .5], [.5, 0]])]
    blocks = []
    for cr in child_ref:
        pts = nodes_of(cr[None])              # child nodes in parent reference coords
        Vc, _, _ = _tri_basis(pts)
        blocks.append(Vc @ C)                 # (7 x 7) parent nodal values -> child nodal values
    Pblk = np.vstack(blocks)                  # 28 x 7
    Pblk[np.abs(Pblk) < 1e-14] = 0.0
    inj = np.zeros((7, 28))
    # parent v1,v2,v3 = child0.v1, child1.v2, child2.v3 ; m12 = child0.v2 ; m23 = child1.v3 ;
    # m31 = child0.v3 ; centroid = centroid of the middle child (child 3)
    for prow, frow in enumerate([0, 7 + 1, 14 + 2, 1, 7 + 2, 2, 21 + 6]):
        inj[prow, frow] = 1.0
    refine, coarsen = [], []
    for l in range(1, L):
        nel = tris[l - 1].shape[0]
        refine.append(_bdiag(np.broadcast_to(Pblk, (nel, 28, 7))))
        coarsen.append(_bdiag(np.broadcast_to(inj, (nel, 7, 28))))
    refine.append(sp.identity(n, format="csr"))
    coarsen.append(sp.identity(n, format="csr"))
    full, dirichlet = [], []
    for l in range(1, L + 1):
        xl = nodes_of(tris[l - 1])
        key = np.round(xl * (3 * 2 ** (L + 8))).astype(np.int64)
        uniq, inv, counts = np.unique(key, axis=0, return_inverse=True, return_counts=True)
        inv = inv.reshape(-1)
        nl = xl.shape[0]
        S = sp.csr_matrix((np.ones(nl), (np.arange(nl), inv)), shape=(nl, uniq.shape[0]))
        # boundary: an edge midpoint owned by exactly one element, its end vertices
        loc = np.arange(nl) % 7
        is_mid = (loc >= 3) & (loc <= 5)
        bmid = is_mid & (counts[inv] == 1)
        bnd = np.zeros(uniq.shape[0], dtype=bool)
        bnd[inv[bmid]] = True
        el = np.arange(nl) // 7
        ends = {3: (0, 1), 4: (1, 2), 5: (2, 0)}
        for r in np.nonzero(bmid)[0]:
            for vv in ends[int(loc[r])]:
                bnd[inv[el[r] * 7 + vv]] = True
        P = S
        for k in range(l - 1, L - 1):
            P = refine[k] @ P
        P = sp.csr_matrix(P)
        P.eliminate_zeros()
        full.append(P)
        dirichlet.append(sp.csr_matrix(P[:, np.nonzero(~bnd)[0]]))
    return Geometry(dict(kind="fem2d", L=L, dim=2, block=7, K=K), x, w,
                    dict(full=full, dirichlet=dirichlet), dict(id=ident, dx=dx, dy=dy), refine, coarsen)


def _lagrange_1d(k):
    """Equispaced nodes on [0,1], values/derivative matrices of the Lagrange basis, Newton-Cotes weights."""
    xi = np.linspace(0.0, 1.0, k + 1)
    V = np.vander(xi, k + 1, increasing=True)
    C = np.linalg.inv(V)                                   # monomial coefficients of the nodal basis

    def basis(pts):
        return np.vander(np.asarray(pts, dtype=np.float64), k + 1, increasing=True) @ C

    def dbasis(pts):
        pts = np.asarray(pts, dtype=np.float64)
        dV = np.zeros((pts.size, k + 1))
        for m in range(1, k + 1):
            dV[:, m] = m * pts ** (m - 1)
        return dV @ C

    wq = {1: [1 / 2, 1 / 2], 2: [1 / 6, 4 / 6, 1 / 6], 3: [1 / 8, 3 / 8, 3 / 8, 1 / 8]}[k]
    return xi, basis, dbasis, np.array(wq)


def fem3d(L: int = 2, k: int = 3) -> Geometry:
    """3-D broken Q_k hexahedra ((k+1)^3 nodes per element, k = 1..3) on [-1,1]^3, octree refinement:
    8^(l-1) elements at level l (reference: fem3d_mpi, src:696-702, "Q_k", k=3 default src:682-684).
    Equispaced tensor nodes (x fastest), Newton-Cotes tensor quadrature at the nodes, children of an element
    in the order (cx + 2 cy + 4 cz)."""
    if k not in (1, 2, 3):
        raise ValueError("fem3d: k must be 1, 2 or 3")
    xi, basis, dbasis, wq = _lagrange_1d(k)
    m1 = k + 1
    nloc = m1 ** 3

    def elements(l):
        """(ne, 3) integer lower corners in units of the level-l cell, in refinement order."""
        E = np.zeros((1, 3), dtype=np.int64)
        for _ in range(1, l):
            off = np.array([[cx, cy, cz] for cz in (0, 1) for cy in (0, 1) for cx in (0, 1)], dtype=np.int64)
            E = (2 * E[:, None, :] + off[None, :, :]).reshape(-1, 3)
        return E

    loc = np.array([[i, j, m] for m in range(m1) for j in range(m1) for i in range(m1)], dtype=np.int64)
    E = elements(L)
    ne = E.shape[0]
    ncell = 2 ** (L - 1)
    h = 2.0 / ncell
    x = (-1.0 + h * (E[:, None, :] + xi[loc][None, :, :])).reshape(-1, 3)
    w3 = (wq[loc[:, 0]] * wq[loc[:, 1]] * wq[loc[:, 2]]) * h ** 3
    w = np.tile(w3, ne)
    n = ne * nloc
    B, dB = basis(xi), dbasis(xi) / h * 1.0                  # d/dx = (1/h) d/dxi
    I1 = np.eye(m1)
    Dx = np.kron(I1, np.kron(I1, dB))                          # x fastest
    Dy = np.kron(I1, np.kron(dB, I1))
    Dz = np.kron(dB, np.kron(I1, I1))
    for M_ in (Dx, Dy, Dz):
        M_[np.abs(M_) < 1e-13 / h] = 0.0
    ops = {"id": sp.identity(n, format="csr")}
    for key, blk in (("dx", Dx), ("dy", Dy), ("dz", Dz)):
        ops[key] = _bdiag(np.broadcast_to(blk, (ne, nloc, nloc)))
    # refine: parent nodal values -> the 8 children's nodal values
    P1 = [basis(xi / 2), basis(0.5 + xi / 2)]                 # child 0 / 1 along one axis
    blocks = []
    for cz in (0, 1):
        for cy in (0, 1):
            for cx in (0, 1):
                blocks.append(np.kron(P1[cz], np.kron(P1[cy], P1[cx])))
    Pblk = np.vstack(blocks)
    Pblk[np.abs(Pblk) < 1e-14] = 0.0
    # coarsen by injection: parent node (i,j,m) sits at child node (2i mod k ...) of the child containing it
    inj = np.zeros((nloc, 8 * nloc))
    for r, (i, j, m) in enumerate(loc):
        c, q = [], []
        for a in (i, j, m):
            t2 = 2 * a                                        # position in units of h_child/k
            ca = 1 if t2 > k else 0                           # ties (t2 == k) go to child 0
            c.append(ca)
            q.append(t2 - ca * k)
        child = c[0] + 2 * c[1] + 4 * c[2]
        inj[r, child * nloc + q[0] + m1 * (q[1] + m1 * q[2])] = 1.0
    refine, coarsen = [], []
    for l in range(1, L):
        nel = 8 ** (l - 1)
        refine.append(_bdiag(np.broadcast_to(Pblk, (nel, 8 * nloc, nloc))))
        coarsen.append(_bdiag(np.broadcast_to(inj, (nel, nloc, 8 * nloc))))
    refine.append(sp.identity(n, format="csr"))
    coarsen.append(sp.identity(n, format="csr"))
    full, dirichlet = [], []
    for l in range(1, L + 1):
        El = elements(l)
        nc = 2 ** (l - 1)
        G = k * El[:, None, :] + loc[None, :, :]              # global tensor-grid index of every broken node
        G = G.reshape(-1, 3)
        npts = k * nc + 1
        gid = G[:, 0] + npts * (G[:, 1] + npts * G[:, 2])
        nl = gid.size
        S = sp.csr_matrix((np.ones(nl), (np.arange(nl), gid)), shape=(nl, npts ** 3))
        interior = np.all((G > 0) & (G < npts - 1), axis=1)
        keep = np.zeros(npts ** 3, dtype=bool)
        keep[gid[interior]] = True
        Pm = S
        for kk in range(l - 1, L - 1):
            Pm = refine[kk] @ Pm
        Pm = sp.csr_matrix(Pm)
        Pm.eliminate_zeros()
        full.append(Pm)
        dirichlet.append(sp.csr_matrix(Pm[:, np.nonzero(keep)[0]]))
    return Geometry(dict(kind="fem3d", L=L, dim=3, block=nloc, k=k), x, w,
                    dict(full=full, dirichlet=dirichlet), ops, refine, coarsen)


def _bdiag(blocks):
    """Block-diagonal CSR from an (ne, r, c) array of dense blocks."""
    blocks = np.asarray(blocks)
    ne, r, c = blocks.shape
    rows = (np.arange(ne)[:, None, None] * r + np.arange(r)[None, :, None]) + np.zeros((1, 1, c), dtype=np.int64)
    cols = (np.arange(ne)[:, None, None] * c + np.arange(c)[None, None, :]) + np.zeros((1, r, 1), dtype=np.int64)
    M = sp.csr_matrix((blocks.reshape(-1), (rows.reshape(-1), cols.reshape(-1))), shape=(ne * r, ne * c))
    M.eliminate_zeros()
    return M


# ----------------------------------------------------------------------------
# AMG hierarchy (upstream `amg`; layout pinned by test/test_d0_construction.jl:82-100)
# ----------------------------------------------------------------------------


@dataclass
class AMG:
    geometry: Geometry
    x: np.ndarray
    w: np.ndarray
    R: List[sp.csr_matrix]        # R[l] = blockdiag(subspaces[sv_k][l] for k)   (S*n x N_l)
    D: List[sp.csr_matrix]        # D[k] = hcat(Z.., operators[op_k], ..Z)      (n x S*n)
    state_variables: Sequence[Sequence[str]]
    Dspec: Sequence[Sequence[str]]


DEFAULT_STATE = (("u", "dirichlet"), ("s", "full"))
DEFAULT_D = {1: (("u", "id"), ("u", "dx"), ("s", "id")),
             2: (("u", "id"), ("u", "dx"), ("u", "dy"), ("s", "id")),
             3: (("u", "id"), ("u", "dx"), ("u", "dy"), ("u", "dz"), ("s", "id"))}   # src:736
DEFAULT_F = {1: lambda x: np.array([0.5, 0.0, 1.0]), 2: lambda x: np.array([0.5, 0.0, 0.0, 1.0]),
             3: lambda x: np.array([0.5, 0.0, 0.0, 0.0, 1.0])}                                  # src:737
DEFAULT_G = {1: lambda x: np.array([x[0], 2.0]), 2: lambda x: np.array([x[0] ** 2 + x[1] ** 2, 100.0]),
             3: lambda x: np.array([x[0] ** 2 + x[1] ** 2 + x[2] ** 2, 100.0])}                 # src:738


def amg(geometry: Geometry, state_variables=DEFAULT_STATE, D=None) -> AMG:
    dim = geometry.discretization["dim"]
    D = DEFAULT_D[dim] if D is None else D
    n = geometry.x.shape[0]
    L = len(geometry.refine)
    names = [sv[0] for sv in state_variables]
    # "fixed": a state variable without unknowns (n x 0 block) -- data the barrier reads through a row of D, e.g. an
    # x-dependent obstacle; not a key of geometry.subspaces (the reference's geometries have :dirichlet / :full only)
    blk = lambda name, l: sp.csr_matrix((n, 0)) if name == "fixed" else geometry.subspaces[name][l]
    R = [sp.csr_matrix(sp.block_diag([blk(sv[1], l) for sv in state_variables], format="csr")) for l in range(L)]
    Z = sp.csr_matrix((n, n))
    Dm = []
    for var, op in D:
        foo = [Z] * len(names)
        foo[names.index(var)] = geometry.operators[op]
        Dm.append(sp.hstack(foo, format="csr"))
    return AMG(geometry, geometry.x, geometry.w, R, Dm, state_variables, D)


# ----------------------------------------------------------------------------
# Convex set + barrier (upstream `convex_Euclidian_power`, `barrier`)
# ----------------------------------------------------------------------------


def barrier_mu(p):
    """log-s multiplicity of the power-cone barrier; elementwise for per-node exponents."""
    if np.isscalar(p):
        return 0.0 if p == 2 else (1.0 if p < 2 else 2.0)
    return np.where(p == 2, 0.0, np.where(p < 2, 1.0, 2.0))


@dataclass
class PowerConeBarrier:
    """Q = {(q, s): s >= |q|_2^p}; F = -log(s^(2/p) - |q|^2) - mu(p) log s, acting on
    Dz[:, idx] = (q_1..q_d, s).  Vectorised over rows; closed-form F1/F2 (the Julia
    default is ForwardDiff of F: identical up to rounding)."""
    idx: Sequence[int]
    p: float
    idx_s2: int = -1     # feasibility phase: the cone's slack is Y[:, idx[-1]] + Y[:, idx_s2]

    def _qs(self, Y):
        s = Y[:, self.idx[-1]]
        if self.idx_s2 >= 0:
            s = s + Y[:, self.idx_s2]
        return Y[:, list(self.idx[:-1])], s

    def phi(self, Y):
        """Distance function of the cone: phi = s^(2/p) - |q|^2 (negative when s <= 0)."""
        q, s = self._qs(Y)
        a = 2.0 / self.p
        with np.errstate(all="ignore"):
            return np.where(s > 0, np.power(np.abs(s), a), -1.0) - np.sum(q * q, axis=1)

    def F(self, x, Y):
        s = self._qs(Y)[1]
        phi = self.phi(Y)
        with np.errstate(all="ignore"):
            val = -np.log(phi) - barrier_mu(self.p) * np.log(s)
            val = np.where((phi > 0) & (s > 0), val, np.inf)
        return val

    def F1(self, x, Y):
        n, K = Y.shape
        q, s = self._qs(Y)
        a = 2.0 / self.p
        mu = barrier_mu(self.p)
        phi = np.power(s, a) - np.sum(q * q, axis=1)
        ds = a * np.power(s, a - 1)
        G = np.zeros((n, K))
        G[:, list(self.idx[:-1])] = 2 * q / phi[:, None]
        G[:, self.idx[-1]] = -ds / phi - mu / s
        if self.idx_s2 >= 0:
            G[:, self.idx_s2] = G[:, self.idx[-1]]
        return G

    def F2(self, x, Y):
        n, K = Y.shape
        qi, si = list(self.idx[:-1]), self.idx[-1]
        q, s = self._qs(Y)
        a = 2.0 / self.p
        mu = barrier_mu(self.p)
        phi = np.power(s, a) - np.sum(q * q, axis=1)
        ds = a * np.power(s, a - 1)
        dds = a * (a - 1) * np.power(s, a - 2)
        H = np.zeros((n, K, K))
        for i, ci in enumerate(qi):
            for j, cj in enumerate(qi):
                H[:, ci, cj] = 4 * q[:, i] * q[:, j] / phi ** 2 + (2 / phi if i == j else 0.0)
            H[:, ci, si] = H[:, si, ci] = -2 * q[:, i] * ds / phi ** 2
        H[:, si, si] = -dds / phi + ds * ds / phi ** 2 + mu / (s * s)
        if self.idx_s2 >= 0:          # the extra slack column repeats the s row/column
            s2 = self.idx_s2
            H[:, s2, :] = H[:, si, :]
            H[:, :, s2] = H[:, :, si]
            H[:, s2, s2] = H[:, si, si]
        return H


def convex_Euclidian_power(idx, p) -> PowerConeBarrier:
    """p: the exponent, or an array of per-node exponents p(x_q) (upstream accepts a function p(x); callers evaluate it at the
    nodes): every formula above is elementwise in p."""
    return PowerConeBarrier(tuple(idx), float(p) if np.isscalar(p) else np.asarray(p, dtype=np.float64))


@dataclass
class LinearBarrier:
    """Half space {y: sum_i coef[i] y[idx[i]] + off > 0} (upstream `convex_linear` with one constant row: bounds, constant
    obstacles): F = -log(coef . y[idx] + off); the cone distance phi of the line search is that affine form."""
    idx: Sequence[int]
    coef: Sequence[float]
    off: float

    def phi(self, Y):
        return Y[:, list(self.idx)] @ np.asarray(self.coef, dtype=np.float64) + self.off

    def F(self, x, Y):
        phi = self.phi(Y)
        with np.errstate(all="ignore"):
            return np.where(phi > 0, -np.log(phi), np.inf)

    def F1(self, x, Y):
        G = np.zeros_like(Y)
        phi = self.phi(Y)
        for i, ci in zip(self.idx, self.coef):
            G[:, i] = -ci / phi
        return G

    def F2(self, x, Y):
        n, K = Y.shape
        H = np.zeros((n, K, K))
        phi = self.phi(Y)
        for i, ci in zip(self.idx, self.coef):
            for j, cj in zip(self.idx, self.coef):
                H[:, i, j] = ci * cj / phi ** 2
        return H


@dataclass
class ConeIntersection:
    """Intersection of power cones acting on disjoint column sets of Dz (upstream `intersect` of
    convex sets, used by parabolic_solve): the barrier is the sum of the cone barriers."""
    cones: Sequence[PowerConeBarrier]

    def phi(self, Y):
        return np.stack([Q.phi(Y) for Q in self.cones], axis=1)

    def F(self, x, Y):
        return sum(Q.F(x, Y) for Q in self.cones)

    def F1(self, x, Y):
        return sum(Q.F1(x, Y) for Q in self.cones)

    def F2(self, x, Y):
        return sum(Q.F2(x, Y) for Q in self.cones)


@dataclass
class ConvexPiecewise:
    """upstream `convex_piecewise` [UPSTREAM-UNVERIFIED recollection: a convex set that varies in space -- at x the intersection of
    the pieces Q[i] with select(x)[i] true]: the barrier at node q is the sum of the barriers of the pieces active there; an
    inactive piece constrains nothing at that node (distance +inf).  `mask`: n x len(cones) booleans (select evaluated at the
    nodes)."""
    cones: Sequence[PowerConeBarrier]
    mask: np.ndarray

    def phi(self, Y):
        return np.stack([np.where(self.mask[:, i], Q.phi(Y), np.inf) for i, Q in enumerate(self.cones)], axis=1)

    def F(self, x, Y):
        with np.errstate(all="ignore"):
            return sum(np.where(self.mask[:, i], Q.F(x, Y), 0.0) for i, Q in enumerate(self.cones))

    def F1(self, x, Y):
        with np.errstate(all="ignore"):
            return sum(np.where(self.mask[:, i, None], Q.F1(x, Y), 0.0) for i, Q in enumerate(self.cones))

    def F2(self, x, Y):
        with np.errstate(all="ignore"):
            return sum(np.where(self.mask[:, i, None, None], Q.F2(x, Y), 0.0) for i, Q in enumerate(self.cones))


def convex_piecewise(cones, select, x) -> ConvexPiecewise:
    """select(x_q) -> one boolean per piece; evaluated at the nodes x (n x dim)."""
    mask = np.array([[bool(b) for b in select(xi)] for xi in x], dtype=bool)
    if mask.shape != (x.shape[0], len(cones)) or not mask.any(axis=1).all():
        raise ValueError("convex_piecewise: select(x) must give one flag per piece and keep at least one piece at every node")
    return ConvexPiecewise(list(cones), mask)


class Barrier:
    """upstream `barrier(F)` -> (f0, f1, f2); algebra pinned by
    test/test_apply_d.jl:44 (apply_D), test/test_column_extract.jl:50-80 (f1 pieces) and
    test/test_map_rows_compare.jl:102-123,165-170 (f2 + restriction)."""

    def __init__(self, Q: PowerConeBarrier):
        self.Q = Q

    @staticmethod
    def apply_D(D, z):
        return np.stack([Dk @ z for Dk in D], axis=1)

    def _Dz(self, s, R, D, z0, pre):
        """Dz at z0 + R s.  `pre = (Dz0, BR)` with Dz0 = D z0 (n x K) and BR[k] = D_k R evaluates
        Dz0 + [BR_k s]_k instead: the driver carries Dz0 forward as the accepted Dz of the previous
        Newton solve, so the start of the next solve sees bit-for-bit the values already verified to be
        inside the cone (re-evaluating D(z0 + R s) has cancellation noise ~1e-13 in dx*u, enough to flip
        the sign of phi = s^2 - |grad u|^2 on plateaus where both vanish like 1/t)."""
        if pre is None:
            return self.apply_D(D, z0 + R @ s)
        Dz0, BR = pre
        return Dz0 + np.stack([BRk @ s for BRk in BR], axis=1)

    def f0(self, s, x, w, c, R, D, z0, pre=None):
        return self.f0_phi(s, x, w, c, R, D, z0, pre=pre)[0]

    def f0_phi(self, s, x, w, c, R, D, z0, phi_ref=None, pre=None):
        """(objective, per-row phi).  With `phi_ref` (phi at the current iterate) the trial is also
        rejected (objective = inf) unless phi >= FRAC_TO_BOUNDARY * phi_ref in every row."""
        Dz = self._Dz(s, R, D, z0, pre)
        phi = self.Q.phi(Dz)
        y = self.Q.F(x, Dz)
        if not np.all(np.isfinite(y)):
            return np.inf, phi
        if phi_ref is not None and np.any(phi < FRAC_TO_BOUNDARY * phi_ref):
            return np.inf, phi
        return float(np.dot(w, y) + sum(np.dot(w * c[:, k], Dz[:, k]) for k in range(len(D)))), phi

    def f1(self, s, x, w, c, R, D, z0, pre=None):
        Dz = self._Dz(s, R, D, z0, pre)
        y = self.Q.F1(x, Dz) + c
        ret = np.zeros(D[0].shape[1])
        for k in range(len(D)):
            ret += D[k].T @ (w * y[:, k])
        return R.T @ ret

    def f2(self, s, x, w, c, R, D, z0, pre=None):
        Dz = self._Dz(s, R, D, z0, pre)
        y = self.Q.F2(x, Dz)
        return hessian_recipe(D, w, y, R)


def hessian_recipe(D, w, y, R=None):
    """H = sum_j D_j' diag(w.y_jj) D_j + sum_{k<j} (D_j' diag(w.y_jk) D_k + D_k' diag(w.y_jk) D_j); R'HR.
    Literal restatement of test/test_map_rows_compare.jl:111-122,170."""
    nD = len(D)
    m0 = D[0].shape[1]
    ret = sp.csr_matrix((m0, m0))
    for j in range(nD):
        foo = amgb_diag(w * y[:, j, j])
        ret = ret + D[j].T @ foo @ D[j]
        for k in range(j):
            if not np.any(y[:, j, k]):
                continue
            foo = amgb_diag(w * y[:, j, k])
            ret = ret + D[j].T @ foo @ D[k] + D[k].T @ foo @ D[j]
    return sp.csr_matrix(R.T @ ret @ R) if R is not None else sp.csr_matrix(ret)


# ----------------------------------------------------------------------------
# Newton / line search / level loop / t-continuation (upstream newton, amgb_step, amgb_core)
# ----------------------------------------------------------------------------

BETA = 0.5          # backtracking factor
ARMIJO = 0.1        # sufficient-decrease constant
MIN_STEP = 1e-8     # give up the line search below this step length
INITIAL_CENTERING_ATTEMPTS = 8   # Newton budgets allowed for the first centering at t0
KAPPA_GROW_FRAC = 0.25  # kappa grows back (kappa <- min(kappa0, kappa^2)) only after a centering with <= 25 % of max_newton
REFINE = True       # after the first acceptable step keep halving while the objective improves
FRAC_TO_BOUNDARY = 0.1  # a step may not shrink any row's cone distance phi below this fraction of its value


def solve(H, g):
    """upstream `MultiGridBarrier.solve(A,b) = A \\ b` (test/test_instrumented_solve.jl:25-28,99)."""
    H = sp.csc_matrix(H)
    if H.shape[0] == 0:
        return np.zeros(0)
    return spla.splu(H).solve(g)


def linesearch_backtracking(x, y, g, n, inc, F0, F1, phi=None):
    """Backtracking on s in {1, beta, beta^2, ...}: the trial must be finite
    (`amgb_all_isfinite`, src:121), keep every row at least FRAC_TO_BOUNDARY of its previous distance
    to the cone boundary (the w-weighted log barrier alone lets Armijo accept points that sit on the
    boundary up to rounding) and satisfy y(x - s n) <= y - ARMIJO*s*inc.  F0(x, phi_ref) -> (y, phi)."""
    s = 1.0
    while s >= MIN_STEP:
        xn = x - s * n
        yn, phin = F0(xn, phi)
        if math.isfinite(yn) and yn <= y - ARMIJO * s * inc:
            # do not overshoot the 1-D minimum towards the boundary: keep halving while the objective
            # still improves (grid version of the exact line search)
            while REFINE and s * BETA >= MIN_STEP:
                x2 = x - (s * BETA) * n
                y2, phi2 = F0(x2, phi)
                if not (math.isfinite(y2) and y2 < yn):
                    break
                xn, yn, phin, s = x2, y2, phi2, s * BETA
            gn = F1(xn)
            if amgb_all_isfinite(gn):
                return xn, yn, gn, s, phin
        s *= BETA
    return x, y, g, 0.0, phi


def stopping_exact(theta):
    return lambda ymin, ynext, gmin, gnext, incmin, inc: ynext >= ymin and gnext >= theta * gmin


def stopping_inexact(lam_tol, theta):
    ex = stopping_exact(theta)
    return lambda ymin, ynext, gmin, gnext, incmin, inc: inc < lam_tol or ex(ymin, ynext, gmin, gnext, incmin, inc)


def newton(F0, F1, F2, x, maxit, stopping_criterion, log=None):
    y, phi = F0(x, None)
    assert math.isfinite(y), "newton: infeasible start"
    g = F1(x)
    ymin, gmin, incmin = y, float(np.linalg.norm(g)), math.inf
    k, converged = 0, False
    while k < maxit and not converged:
        k += 1
        H = F2(x)
        nstep = solve(H, g)
        if not amgb_all_isfinite(nstep):
            break
        inc = float(np.dot(g, nstep))
        if inc <= 0:
            converged = True
            break
        xn, yn, gn, s, phi = linesearch_backtracking(x, y, g, nstep, inc, F0, F1, phi)
        gnn = float(np.linalg.norm(gn))
        if stopping_criterion(ymin, yn, gmin, gnn, incmin, inc):
            converged = True
        x, y, g = xn, yn, gn
        ymin, gmin, incmin = min(ymin, y), min(gmin, gnn), min(incmin, inc)
        if log is not None:
            log.append(dict(k=k, y=y, gnorm=gnn, inc=inc, s=s))
    return dict(x=x, y=y, k=k, converged=converged)


LEVEL_SCHEDULE = "fine"   # "fine": Newton on the finest subspace only; "all": coarse -> fine level loop


def level_schedule(L, schedule=None):
    schedule = LEVEL_SCHEDULE if schedule is None else schedule
    return [L - 1] if schedule == "fine" else list(range(L))


# Newton's stopping rule on the finest level at the intermediate values of t: "exact" (default) = stagnation of the objective at
# every t; "decrement" = stop as soon as the Newton decrement <g, n> falls below DECREMENT_FRAC min w (the path is followed, not
# resolved) and keep the stagnation rule for the LAST t, whose centre is the answer.  Both end at the same point (<= 5e-12).
# Measured (profiles/r3_centering_counts.txt, Newton steps exact / decrement): fem2d L=7 p=1.5 88 / 75, fem3d L=4 104 / 86, but
# fem2d L=7 p=1 466 / 580 and L=8 p=1 668 / 677 -- at p = 1 the steps go into damped progress towards each centre, not into
# resolving it, and starting the next t from a looser point costs more than it saves: hence the default.  [UPSTREAM-UNVERIFIED]
CENTERING = "exact"
# threshold of the decrement rule on the finest level, as a fraction of min w: w F is self-concordant only after division by w, so
# "Newton converges quadratically from here" means <g, n> / min w below a small constant.  (sqrt(min w) / 2, the coarse levels'
# lam_tol, is far too loose here: at fem2d L=4, p=1 the decrement idles at 0.0187 < 0.0197 for hundreds of full steps.)
DECREMENT_FRAC = 0.01


def amgb_step(B: Barrier, M: AMG, z, Dz0, c, maxit, lam_tol, log=None, schedule=None, final=True):
    """One centering at fixed t (c already scaled by t): Newton on the subspaces R[J] for J in the level
    schedule, each from s = 0, z += R[J] s.  its[l] = Newton steps on level l (docs/src/guide.md:158
    `sum(SOL_main.its)`).  The literal coarse->fine loop ("all") de-centres the iterate at large t (coarse
    directions drive single rows into the cone boundary) and costs ~8x more Newton steps than plain
    path-following on the finest level, which is therefore the default."""
    L = len(M.R)
    its = np.zeros(L, dtype=np.int64)
    converged = True
    if not hasattr(M, "BR"):
        M.BR = {}
    for J in level_schedule(L, schedule):
        R = M.R[J]
        if J not in M.BR:
            M.BR[J] = [sp.csr_matrix(Dk @ R) for Dk in M.D]
        pre = (Dz0, M.BR[J])
        s0 = np.zeros(R.shape[1])
        if J == L - 1:
            crit = stopping_exact(0.1) if (final or CENTERING == "exact") else stopping_inexact(DECREMENT_FRAC * float(np.min(M.w)), 0.1)
        else:
            crit = stopping_inexact(lam_tol, 0.5)
        lg = [] if log is not None else None
        SOL = newton(lambda s, ref: B.f0_phi(s, M.x, M.w, c, R, M.D, z, ref, pre),
                     lambda s: B.f1(s, M.x, M.w, c, R, M.D, z, pre),
                     lambda s: B.f2(s, M.x, M.w, c, R, M.D, z, pre),
                     s0, maxit, crit, lg)
        its[J] = SOL["k"]
        if log is not None:
            log.append(dict(level=J, newton=lg))
        z = z + R @ SOL["x"]
        Dz0 = B._Dz(SOL["x"], R, M.D, z, pre)          # the accepted Dz, carried forward
        if J == L - 1:
            converged = SOL["converged"]
    return dict(z=z, Dz0=Dz0, its=its, converged=converged)


STOP_RULE = "fixed"      # "fixed": the continuation ends at t_stop (below); "upstream": the literal `while t <= 1/tol`, t <- kappa t


def amgb_core(B: Barrier, M: AMG, z, c, tol, t=0.1, maxit=10000, kappa=10.0, max_newton=None, log=None,
              schedule=None, early_stop=None, stop_rule=None):
    """`early_stop(Dz0) -> bool` is evaluated after every centering (feasibility phase: stop as soon as the
    original cone is strictly satisfied).  `stop_rule`: see STOP_RULE; both rules are [UPSTREAM-UNVERIFIED] (SURVEY.md
    Appendix A recalls `t <- kappa t until t > 1/tol`); they visit the same ts whenever kappa is never reduced."""
    stop_rule = STOP_RULE if stop_rule is None else stop_rule
    if stop_rule not in ("fixed", "upstream"):
        raise ValueError("stop_rule must be 'fixed' or 'upstream'")
    if max_newton is None:
        max_newton = int(math.ceil(math.log2(-math.log2(np.finfo(np.float64).eps)))) + 2 + 40
    lam_tol = math.sqrt(float(np.min(M.w))) / 2
    t_begin = time.time()
    kappa0 = kappa
    its, ts, cdots = [], [], []

    def cdot(Dz):
        return float(sum(np.dot(M.w * c[:, k], Dz[:, k]) for k in range(len(M.D))))

    Dz0 = B.apply_D(M.D, z)
    # initial centering: a far-away start (e.g. the previous time step of parabolic_solve, which sits next to
    # the cone boundary) may need more than one Newton budget; keep centering from the improved iterate
    it0 = np.zeros(len(M.R), dtype=np.int64)
    t_stop = t
    while t_stop <= 1 / tol:
        t_stop *= kappa0
    fixed = stop_rule == "fixed"
    going = (lambda tt: tt < t_stop) if fixed else (lambda tt: tt <= 1 / tol)
    for attempt in range(INITIAL_CENTERING_ATTEMPTS):
        SOL = amgb_step(B, M, z, Dz0, t * c, max_newton, lam_tol, log, schedule, final=(early_stop is not None) or not going(t))
        it0 += SOL["its"]
        z, Dz0 = SOL["z"], SOL["Dz0"]
        if SOL["converged"]:
            break
    else:
        raise RuntimeError("amgb: initial centering failed at t=%g" % t)
    its.append(it0); ts.append(t); cdots.append(cdot(Dz0))
    k = 1
    stopped = early_stop is not None and early_stop(Dz0)
    # The continuation ends at a FIXED barrier parameter: t_stop = the first value of the nominal sequence t0 * kappa0^k beyond
    # 1 / tol (1e8 for the defaults).  Without it the last t depends on the history of kappa reductions -- a discrete, rounding
    # sensitive path -- and two correct runs end at different central points (fem2d L=7, p=1: t = 1.0e8 or 1.8e8, z 1e-6 apart).
    # stop_rule = "upstream" keeps the literal loop: continue while t <= 1 / tol, every step t <- kappa t.
    while going(t) and kappa > 1 and k < maxit and not stopped:
        k += 1
        it_k = np.zeros(len(M.R), dtype=np.int64)
        while kappa > 1:
            t1 = min(kappa * t, t_stop) if fixed else kappa * t
            fin = (early_stop is not None) or not going(t1)      # phases that may stop at any centering resolve every one
            SOL = amgb_step(B, M, z, Dz0, t1 * c, max_newton, lam_tol, log, schedule, final=fin)
            it_k += SOL["its"]
            if fin and early_stop is None and CENTERING == "decrement":
                # the last centre is resolved from a point that was only followed: it may take more than one Newton budget, as the
                # initial centering may (keep centering from the improved iterate before giving kappa up)
                for attempt in range(INITIAL_CENTERING_ATTEMPTS - 1):
                    if SOL["converged"]:
                        break
                    SOL = amgb_step(B, M, SOL["z"], SOL["Dz0"], t1 * c, max_newton, lam_tol, log, schedule, final=True)
                    it_k += SOL["its"]
            if SOL["converged"]:
                if SOL["its"].max() <= max_newton * KAPPA_GROW_FRAC:
                    kappa = min(kappa0, kappa * kappa)
                z, Dz0, t = SOL["z"], SOL["Dz0"], t1
                break
            kappa = math.sqrt(kappa)
            if kappa < 1 + 1e-3:
                kappa = 1.0
        its.append(it_k); ts.append(t); cdots.append(cdot(Dz0))
        stopped = early_stop is not None and early_stop(Dz0)
    if going(t) and not stopped:
        raise RuntimeError("amgb: convergence failure at t=%g kappa=%g" % (t, kappa))
    return dict(z=z, its=np.array(its).T, ts=np.array(ts), c_dot_Dz=np.array(cdots),
                t_elapsed=time.time() - t_begin)


@dataclass
class AMGBSOL:
    """src:467-473 field order."""
    z: np.ndarray
    SOL_feasibility: Optional[dict]
    SOL_main: dict
    log: list
    geometry: Geometry


def amgb(geometry: Geometry, p=1.0, state_variables=DEFAULT_STATE, D=None, f=None, g=None,
         tol=None, t=0.1, maxit=10000, kappa=10.0, verbose=False, logfile=None, keep_log=False,
         schedule=None, extra=(), cone_idx=None, stop_rule=None, select=None) -> AMGBSOL:
    """`extra`: further convex sets intersected with the p-Laplace power cone (upstream `intersect`), e.g. LinearBarrier;
    `cone_idx`: the rows (q.., s) of D the power cone acts on (default: the last dim + 1 rows)."""
    dim = geometry.discretization["dim"]
    f = DEFAULT_F[dim] if f is None else f
    g = DEFAULT_G[dim] if g is None else g
    tol = math.sqrt(np.finfo(np.float64).eps) if tol is None else tol
    M = amg(geometry, state_variables, D)
    x = M.x
    z0 = map_rows(lambda xi: g(xi), x)           # (n, S)
    c = map_rows(lambda xi: f(xi), x)            # (n, K)
    nD = len(M.D)
    if callable(p):                              # x-dependent exponent p(x): evaluated at the nodes
        p = np.array([float(p(xi)) for xi in x])
    Q = convex_Euclidian_power(idx=list(cone_idx) if cone_idx is not None else
                               (list(range(1, dim + 2)) if nD == dim + 2 else list(range(nD - dim - 1, nD))), p=p)
    B = Barrier(ConeIntersection([Q, *extra]) if extra else Q)
    if select is not None:                       # convex_piecewise: the pieces (Q, *extra), active where select(x) says
        B = Barrier(convex_piecewise([Q, *extra], select, x))
    zvec = z0.reshape(-1, order="F")
    Dz = B.apply_D(M.D, zvec)
    log = [] if keep_log else None
    SOL_feas = None
    if select is not None:
        if not np.all(np.isfinite(B.Q.F(x, Dz))):
            raise RuntimeError("amgb: a piecewise set needs a strictly feasible start")
    elif extra and not np.all(np.isfinite(B.Q.F(x, Dz))):
        zvec, SOL_feas = amgb_phase1_slack(geometry, state_variables, M.Dspec, Q, extra, zvec, Dz, tol, schedule, c=c)
    elif not np.all(np.isfinite(Q.F(x, Dz))):
        zvec, SOL_feas = amgb_phase1(geometry, state_variables, M.Dspec, Q, zvec, Dz, tol, schedule)
    SOL = amgb_core(B, M, zvec, c, tol, t=t, maxit=maxit, kappa=kappa, log=log, schedule=schedule, stop_rule=stop_rule)
    z = SOL.pop("z").reshape(z0.shape, order="F")
    return AMGBSOL(z, SOL_feas, SOL, log or [], geometry)


def amgb_phase1(geometry, state_variables, D, Q: PowerConeBarrier, zvec, Dz, tol, schedule=None):
    """Feasibility phase (upstream amgb_phase1; SOL_feasibility of src:428-455).  For the power-cone family
    the phase-1 problem "find z with Dz strictly inside Q" has a closed-form solution: the cone's slack row
    is `id` applied to a state variable living in the :full subspace, which contains the constants, so
    shifting that variable by sigma = 1 + max_rows(|q|^p - s) lands strictly inside the cone without
    touching u (a relaxed-cone Newton phase would be singular along (ds, dsigma) = (d, -d)).  Other layouts
    (slack in a space without constants, slack row not an identity) are rejected."""
    n = geometry.x.shape[0]
    var, op = D[Q.idx[-1]]
    names = [sv[0] for sv in state_variables]
    if op != "id" or dict(state_variables)[var] != "full":
        raise NotImplementedError("feasibility phase: the cone's slack must be `id` of a :full state variable")
    q, s = Q._qs(Dz)
    sigma = 1.0 + float(np.max(np.sum(q * q, axis=1) ** (Q.p / 2.0) - s))
    z = zvec.copy()
    k = names.index(var)
    z[k * n:(k + 1) * n] += sigma
    L = len(geometry.refine)
    SOL = dict(shift=sigma, its=np.zeros((L, 0), dtype=np.int64), ts=np.zeros(0), c_dot_Dz=np.zeros(0), t_elapsed=0.0)
    return z, SOL


PHASE1_SLACK_FLOOR = 1.0      # the slack field of the general feasibility phase lives in sigma > -PHASE1_SLACK_FLOOR
PHASE1_PENALTY = 10.0         # ... and costs this many times the largest original cost coefficient


def amgb_phase1_slack(geometry, state_variables, D, Q: PowerConeBarrier, extra, zvec, Dz, tol, schedule=None, c=None):
    """General feasibility phase for the cone intersected with half spaces (upstream amgb_phase1: a barrier solve on the
    convex set relaxed by a slack, stopped as soon as the slack is negative; [UPSTREAM-UNVERIFIED] in its details --
    nothing in the reference pins them -- so this is the phase both this oracle and the HIP path implement):
        state (.., sigma) with sigma in the :full subspace, D' = D + [sigma id], minimise  int c . Dz + M sigma  subject to
        (q, s + sigma) in the power cone,  coef . y + off + sigma > 0  for every half space,  sigma > -1,
    from sigma0 = 1 + the largest violation (strictly feasible), path-followed by amgb_core and stopped after the first
    centering with sigma < 0 everywhere: the point then lies strictly inside the original set.  It is the original problem
    relaxed by the slack with the penalty M = PHASE1_PENALTY max(1, |c|_max) on it ("big M"): the original cost keeps the
    cone slack s bounded (with cost on sigma alone the barrier runs off along s -> infinity), M > c_s makes trading sigma
    for s pay until sigma reaches its floor, and the floor sigma > -1 bounds the problem below."""
    n = geometry.x.shape[0]
    K = Dz.shape[1]
    if len(extra) > 1:
        raise NotImplementedError("feasibility phase: one half space (three barrier terms in all)")
    state1 = tuple(state_variables) + (("sigma", "full"),)
    D1 = tuple(D) + (("sigma", "id"),)
    M1 = amg(geometry, state1, D1)
    Q1 = PowerConeBarrier(tuple(Q.idx), Q.p, idx_s2=K)
    terms = [Q1] + [LinearBarrier(list(e.idx) + [K], list(e.coef) + [1.0], e.off) for e in extra]
    terms.append(LinearBarrier([K], [1.0], PHASE1_SLACK_FLOOR))
    B1 = Barrier(ConeIntersection(terms))
    q, sl = Q._qs(Dz)
    viol = [np.sum(q * q, axis=1) ** (Q.p / 2.0) - sl] + [-e.phi(Dz) for e in extra]
    sigma0 = 1.0 + max(0.0, float(np.max(np.concatenate(viol))))
    z1 = np.concatenate([zvec, np.full(n, sigma0)])
    c1 = np.zeros((n, K + 1))
    if c is not None:
        c1[:, :K] = c
    c1[:, K] = PHASE1_PENALTY * max(1.0, float(np.max(np.abs(c1[:, :K]))))
    SOL = amgb_core(B1, M1, z1, c1, tol, schedule=schedule, early_stop=lambda Dz0: bool(np.max(Dz0[:, K]) < 0.0))
    z = SOL.pop("z")
    if not np.max(z[len(zvec):]) < 0.0:
        raise RuntimeError("amgb: the problem is infeasible (the feasibility phase ended with a non-negative slack)")
    SOL["sigma0"] = sigma0
    return z[:len(zvec)], SOL


@dataclass
class ParabolicSOL:
    """src:512-516 field order: geometry, ts, u (one n x S snapshot per time)."""
    geometry: Geometry
    ts: np.ndarray
    u: list


def parabolic_problem(geometry, p):
    """state variables, D and the cone intersection of the implicit-Euler step of the parabolic p-Laplace flow
        u_t - div(|grad u|^(p-2) grad u) = -f1 :
    minimise  int (1/2h)(s1 - 2 u u_k) + (1/p) s2 + f1 u   s.t.  s1 >= u^2,  s2 >= |grad u|^p."""
    dim = geometry.discretization["dim"]
    ops = ("dx", "dy", "dz")[:dim]
    state = (("u", "dirichlet"), ("s1", "full"), ("s2", "full"))
    D = (("u", "id"),) + tuple(("u", o) for o in ops) + (("s1", "id"), ("s2", "id"))
    K = dim + 3
    cones = [([0, K - 2], 2.0), (list(range(1, dim + 1)) + [K - 1], float(p))]
    return state, D, K, cones, ops


def parabolic_initial(geometry, p, g):
    """z0 = [u0; s1; s2]: u0 = g(x)[0] at every node, constant slacks strictly inside both cones."""
    dim = geometry.discretization["dim"]
    ops = ("dx", "dy", "dz")[:dim]
    x = geometry.x
    u0 = np.array([g(xi)[0] for xi in x], dtype=np.float64)
    grad2 = sum((geometry.operators[o] @ u0) ** 2 for o in ops)
    n = x.shape[0]
    s1 = np.full(n, 1.0 + float(np.max(u0 * u0)))
    s2 = np.full(n, 1.0 + float(np.max(grad2 ** (p / 2.0))))
    return np.concatenate([u0, s1, s2])


def parabolic_cost(n, K, p, h, fgrid, uk):
    c = np.zeros((n, K))
    c[:, 0] = fgrid - uk / h
    c[:, K - 2] = 1.0 / (2.0 * h)
    c[:, K - 1] = 1.0 / p
    return c


def parabolic_solve(geometry: Geometry, h=0.2, t0=0.0, t1=1.0, p=1.0, f1=None, g=None, tol=None, verbose=False,
                    schedule=None) -> ParabolicSOL:
    """upstream `parabolic_solve(geometry; h, t1, p, ...)` (kwargs evidenced at test/test_parabolic.jl:48,
    docs/src/guide.md:367,377): implicit Euler, one barrier solve (amgb main phase) per time step; the
    Dirichlet data is the (time-independent) boundary trace of the initial condition g."""
    dim = geometry.discretization["dim"]
    g = DEFAULT_G[dim] if g is None else g
    f1 = (lambda x: 0.5) if f1 is None else f1
    tol = math.sqrt(np.finfo(np.float64).eps) if tol is None else tol
    state, D, K, cones, ops = parabolic_problem(geometry, p)
    M = amg(geometry, state, D)
    B = Barrier(ConeIntersection([convex_Euclidian_power(idx, pp) for idx, pp in cones]))
    n = M.x.shape[0]
    z = parabolic_initial(geometry, p, g)
    fgrid = np.array([f1(xi) for xi in M.x], dtype=np.float64)
    nsteps = int(round((t1 - t0) / h))
    ts = t0 + h * np.arange(nsteps + 1)
    u = [z.reshape(n, 3, order="F").copy()]
    for _ in range(nsteps):
        c = parabolic_cost(n, K, p, h, fgrid, z[:n])
        z = amgb_core(B, M, z, c, tol, schedule=schedule)["z"]
        u.append(z.reshape(n, 3, order="F").copy())
    return ParabolicSOL(geometry, ts, u)


def fem1d_solve(L=4, **kw):
    return amgb(fem1d(L), **kw)


def fem3d_solve(L=2, k=3, **kw):
    return amgb(fem3d(L, k), **kw)


def fem2d_solve(L=2, K=None, **kw):
    return amgb(fem2d(L, K), **kw)
