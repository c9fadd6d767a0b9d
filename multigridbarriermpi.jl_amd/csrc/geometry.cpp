// Native geometry builders (host, setup-time).  In the reference these are MultiGridBarrier's
// fem1d / fem2d, called at src/MultiGridBarrierMPI.jl:561,628 before native_to_mpi.
// Element conventions (shared with oracle/mgb_oracle.py, documented in DESIGN.md):
//   1-D: element e has nodes (left, right).
//   2-D: local node order v1,v2,v3,m12,m23,m31,centroid; children of a triangle in the order
//        (v1,m12,m31) (m12,v2,m23) (m31,m23,v3) (m23,m31,m12); element 4e+c is child c of e.
// The nodal basis is written from its barycentric closed form (P2 + cubic bubble); topology is
// tracked through vertex/edge ids, so no floating-point coordinate hashing is involved.
#include "geometry.hpp"

#include <array>
#include <cmath>
#include <exception>
#include <thread>
#include <utility>
#include <chrono>
#include <cstdio>
#include <cstdlib>

namespace mgb {

namespace {

// level_job(l) for l = 0 .. L-1, one thread per level (the finest one -- the largest numbering, no prolongation -- on the
// calling thread); an exception of any level is rethrown here after all threads have joined
template <class F>
void run_levels(int L, F&& level_job) {
  std::vector<std::thread> pool;
  std::vector<std::exception_ptr> err(L);
  auto guarded = [&](int l) {
    try {
      level_job(l);
    } catch (...) {
      err[l] = std::current_exception();
    }
  };
  for (int l = 0; l + 1 < L; ++l) pool.emplace_back(guarded, l);
  guarded(L - 1);
  for (auto& th : pool) th.join();
  for (auto& e : err)
    if (e) std::rethrow_exception(e);
}

}  // namespace

GeometryHost fem1d_native(int L) {
  if (L < 1 || L > 24) throw ArgError("fem1d: L out of range");
  GeometryHost g;
  g.dim = 1;
  g.block = 2;
  g.L = L;
  const int ne = 1 << L;
  g.n = 2 * ne;
  const double h = 2.0 / ne;
  g.x.resize(g.n);
  g.w.assign(g.n, h / 2);
  std::vector<Triplet> t;
  for (int e = 0; e < ne; ++e) {
    // identical expression to the oracle's linspace(-1,1,ne+1) is not required: parity tests
    // compare to 1e-14; use the exact dyadic form
    g.x[2 * e] = -1.0 + h * e;
    g.x[2 * e + 1] = -1.0 + h * (e + 1);
    for (int i = 0; i < 2; ++i) {
      t.push_back({2 * e + i, 2 * e, -1.0 / h});
      t.push_back({2 * e + i, 2 * e + 1, 1.0 / h});
    }
  }
  g.operators["dx"] = from_triplets(g.n, g.n, t);
  g.operators["id"] = identity(g.n);
  for (int l = 1; l < L; ++l) {
    const int nel = 1 << l;
    std::vector<Triplet> r, c;
    for (int e = 0; e < nel; ++e) {
      const int f = 4 * e, p = 2 * e;
      r.push_back({f, p, 1.0});
      r.push_back({f + 1, p, 0.5});
      r.push_back({f + 1, p + 1, 0.5});
      r.push_back({f + 2, p, 0.5});
      r.push_back({f + 2, p + 1, 0.5});
      r.push_back({f + 3, p + 1, 1.0});
      c.push_back({p, f, 1.0});
      c.push_back({p + 1, f + 3, 1.0});
    }
    g.refine.push_back(from_triplets(4 * nel, 2 * nel, r));
    g.coarsen.push_back(from_triplets(2 * nel, 4 * nel, c));
  }
  g.refine.push_back(identity(g.n));
  g.coarsen.push_back(identity(g.n));
  auto& full = g.subspaces["full"];
  auto& dir = g.subspaces["dirichlet"];
  for (int l = 1; l <= L; ++l) {
    const int nel = 1 << l;
    std::vector<Triplet> sf, sd;
    for (int r = 0; r < 2 * nel; ++r) {
      int v = (r + 1) / 2;  // vertex id 0..nel
      sf.push_back({r, v, 1.0});
      if (v > 0 && v < nel) sd.push_back({r, v - 1, 1.0});
    }
    Csr F = from_triplets(2 * nel, nel + 1, sf), D = from_triplets(2 * nel, nel - 1, sd);
    for (int k = l - 1; k < L - 1; ++k) {
      F = spgemm(g.refine[k], F, true);
      D = spgemm(g.refine[k], D, true);
    }
    full.push_back(std::move(F));
    dir.push_back(std::move(D));
  }
  return g;
}

namespace {

// P2 + bubble nodal basis at reference point (xi, eta): values and (xi,eta)-gradients.
void tri_basis(double xi, double et, double* val, double* dxi, double* det) {
  const double l[3] = {1 - xi - et, xi, et};
  const double gx[3] = {-1, 1, 0}, gy[3] = {-1, 0, 1};
  const double b = l[0] * l[1] * l[2];
  const double bx = gx[0] * l[1] * l[2] + l[0] * gx[1] * l[2] + l[0] * l[1] * gx[2];
  const double by = gy[0] * l[1] * l[2] + l[0] * gy[1] * l[2] + l[0] * l[1] * gy[2];
  for (int i = 0; i < 3; ++i) {
    val[i] = l[i] * (2 * l[i] - 1) + 3 * b;
    dxi[i] = (4 * l[i] - 1) * gx[i] + 3 * bx;
    det[i] = (4 * l[i] - 1) * gy[i] + 3 * by;
  }
  const int ea[3] = {0, 1, 2}, eb[3] = {1, 2, 0};  // m12, m23, m31
  for (int e = 0; e < 3; ++e) {
    const int i = ea[e], j = eb[e];
    val[3 + e] = 4 * l[i] * l[j] - 12 * b;
    dxi[3 + e] = 4 * (l[i] * gx[j] + l[j] * gx[i]) - 12 * bx;
    det[3 + e] = 4 * (l[i] * gy[j] + l[j] * gy[i]) - 12 * by;
  }
  val[6] = 27 * b;
  dxi[6] = 27 * bx;
  det[6] = 27 * by;
}

const double kRefNodes[7][2] = {{0, 0}, {1, 0}, {0, 1}, {.5, 0}, {.5, .5}, {0, .5}, {1.0 / 3, 1.0 / 3}};
const double kTriW[7] = {1.0 / 20, 1.0 / 20, 1.0 / 20, 2.0 / 15, 2.0 / 15, 2.0 / 15, 9.0 / 20};

struct Tri {
  int v[3];
};

}  // namespace

GeometryHost fem2d_native(int L, const double* K, int nK_rows) {
  if (L < 1 || L > 14) throw ArgError("fem2d: L out of range");
  static const double Kdef[12] = {-1, -1, 1, -1, -1, 1, 1, -1, 1, 1, -1, 1};
  if (!K) {
    K = Kdef;
    nK_rows = 6;
  }
  if (nK_rows % 3 != 0 || nK_rows < 3) throw ArgError("fem2d: K must have 3m rows");
  GeometryHost g;
  g.dim = 2;
  g.block = 7;
  g.L = L;
  auto tnow = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  double tph = tnow();
  const bool vt = std::getenv("MGB_VERBOSE_SETUP") != nullptr;
  auto phase = [&](const char* what) {
    if (vt) std::fprintf(stderr, "[mgb setup] fem2d %-22s %.3f s\n", what, tnow() - tph);
    tph = tnow();
  };
  // ---- level-1 topology: deduplicate vertices by exact coordinates
  std::vector<std::array<double, 2>> vx;
  std::map<std::pair<double, double>, int> vmap;
  std::vector<std::vector<Tri>> tris(L);
  for (int e = 0; e < nK_rows / 3; ++e) {
    Tri t;
    for (int i = 0; i < 3; ++i) {
      std::pair<double, double> key(K[2 * (3 * e + i)], K[2 * (3 * e + i) + 1]);
      auto it = vmap.find(key);
      if (it == vmap.end()) {
        it = vmap.emplace(key, (int)vx.size()).first;
        vx.push_back({key.first, key.second});
      }
      t.v[i] = it->second;
    }
    tris[0].push_back(t);
  }
  // ---- red refinement; mid[l] maps an edge of level l to the vertex id of its midpoint
  using Edge = std::pair<int, int>;
  auto mk = [](int a, int b) { return a < b ? Edge(a, b) : Edge(b, a); };
  std::vector<std::map<Edge, int>> mid(L);
  for (int l = 0; l < L; ++l) {
    // midpoints are created for every level (the finest ones serve as edge-node ids/coords)
    for (const Tri& t : tris[l])
      for (int i = 0; i < 3; ++i) {
        Edge e = mk(t.v[i], t.v[(i + 1) % 3]);
        if (!mid[l].count(e)) {
          mid[l][e] = (int)vx.size();
          vx.push_back({(vx[e.first][0] + vx[e.second][0]) / 2, (vx[e.first][1] + vx[e.second][1]) / 2});
        }
      }
    if (l + 1 < L) {
      tris[l + 1].reserve(4 * tris[l].size());
      for (const Tri& t : tris[l]) {
        int m12 = mid[l][mk(t.v[0], t.v[1])], m23 = mid[l][mk(t.v[1], t.v[2])], m31 = mid[l][mk(t.v[2], t.v[0])];
        tris[l + 1].push_back({{t.v[0], m12, m31}});
        tris[l + 1].push_back({{m12, t.v[1], m23}});
        tris[l + 1].push_back({{m31, m23, t.v[2]}});
        tris[l + 1].push_back({{m23, m31, m12}});
      }
    }
  }
  phase("refinement");
  // ---- finest level: coordinates, weights, dx, dy
  const std::vector<Tri>& T = tris[L - 1];
  const int ne = (int)T.size();
  g.n = 7 * ne;
  g.x.resize((size_t)g.n * 2);
  g.w.resize(g.n);
  double rdxi[7][7], rdet[7][7];  // [node][basis]
  for (int i = 0; i < 7; ++i) {
    double val[7];
    tri_basis(kRefNodes[i][0], kRefNodes[i][1], val, rdxi[i], rdet[i]);
  }
  Csr dx(g.n, g.n), dy(g.n, g.n);
  dx.colidx.reserve((size_t)49 * ne);
  dx.vals.reserve((size_t)49 * ne);
  dy.colidx.reserve((size_t)49 * ne);
  dy.vals.reserve((size_t)49 * ne);
  for (int e = 0; e < ne; ++e) {
    const double* p1 = vx[T[e].v[0]].data();
    const double* p2 = vx[T[e].v[1]].data();
    const double* p3 = vx[T[e].v[2]].data();
    double P[7][2];
    for (int d = 0; d < 2; ++d) {
      P[0][d] = p1[d];
      P[1][d] = p2[d];
      P[2][d] = p3[d];
      P[3][d] = (p1[d] + p2[d]) / 2;
      P[4][d] = (p2[d] + p3[d]) / 2;
      P[5][d] = (p3[d] + p1[d]) / 2;
      P[6][d] = (p1[d] + p2[d] + p3[d]) / 3;
    }
    const double e1x = p2[0] - p1[0], e1y = p2[1] - p1[1], e2x = p3[0] - p1[0], e2y = p3[1] - p1[1];
    const double det = e1x * e2y - e1y * e2x;
    if (det == 0) throw ArgError("fem2d: degenerate triangle");
    const double area = std::fabs(det) / 2;
    const double xix = e2y / det, xiy = -e2x / det, etx = -e1y / det, ety = e1x / det;
    for (int i = 0; i < 7; ++i) {
      const int r = 7 * e + i;
      g.x[2 * r] = P[i][0];
      g.x[2 * r + 1] = P[i][1];
      g.w[r] = area * kTriW[i];
      for (int j = 0; j < 7; ++j) {
        double vx_ = xix * rdxi[i][j] + etx * rdet[i][j];
        double vy_ = xiy * rdxi[i][j] + ety * rdet[i][j];
        if (vx_ != 0.0) {
          dx.colidx.push_back(7 * e + j);
          dx.vals.push_back(vx_);
        }
        if (vy_ != 0.0) {
          dy.colidx.push_back(7 * e + j);
          dy.vals.push_back(vy_);
        }
      }
      dx.rowptr[r + 1] = (int)dx.colidx.size();
      dy.rowptr[r + 1] = (int)dy.colidx.size();
    }
  }
  g.operators["dx"] = std::move(dx);
  g.operators["dy"] = std::move(dy);
  g.operators["id"] = identity(g.n);
  phase("operators");
  // ---- refine / coarsen blocks: parent nodal values -> nodal values of the 4 children
  const double childref[4][3][2] = {{{0, 0}, {.5, 0}, {0, .5}}, {{.5, 0}, {1, 0}, {.5, .5}},
                                    {{0, .5}, {.5, .5}, {0, 1}}, {{.5, .5}, {0, .5}, {.5, 0}}};
  double Pblk[28][7];
  for (int c = 0; c < 4; ++c) {
    const double(*q)[2] = childref[c];
    double pts[7][2];
    for (int d = 0; d < 2; ++d) {
      pts[0][d] = q[0][d];
      pts[1][d] = q[1][d];
      pts[2][d] = q[2][d];
      pts[3][d] = (q[0][d] + q[1][d]) / 2;
      pts[4][d] = (q[1][d] + q[2][d]) / 2;
      pts[5][d] = (q[2][d] + q[0][d]) / 2;
      pts[6][d] = (q[0][d] + q[1][d] + q[2][d]) / 3;
    }
    for (int i = 0; i < 7; ++i) {
      double a[7], b[7];
      tri_basis(pts[i][0], pts[i][1], Pblk[7 * c + i], a, b);
      for (int j = 0; j < 7; ++j)
        if (std::fabs(Pblk[7 * c + i][j]) < 1e-14) Pblk[7 * c + i][j] = 0.0;
    }
  }
  const int injrow[7] = {0, 7 + 1, 14 + 2, 1, 7 + 2, 2, 21 + 6};
  for (int l = 0; l + 1 < L; ++l) {
    const int nel = (int)tris[l].size();
    Csr R(28 * nel, 7 * nel), C(7 * nel, 28 * nel);
    for (int e = 0; e < nel; ++e) {
      for (int i = 0; i < 28; ++i) {
        for (int j = 0; j < 7; ++j)
          if (Pblk[i][j] != 0.0) {
            R.colidx.push_back(7 * e + j);
            R.vals.push_back(Pblk[i][j]);
          }
        R.rowptr[28 * e + i + 1] = (int)R.colidx.size();
      }
      for (int i = 0; i < 7; ++i) {
        C.colidx.push_back(28 * e + injrow[i]);
        C.vals.push_back(1.0);
        C.rowptr[7 * e + i + 1] = (int)C.colidx.size();
      }
    }
    g.refine.push_back(std::move(R));
    g.coarsen.push_back(std::move(C));
  }
  g.refine.push_back(identity(g.n));
  g.coarsen.push_back(identity(g.n));
  phase("refine / coarsen");
  // ---- continuous dofs per level: vertices, edges (keyed by their midpoint vertex id), centroids
  // one thread per level: the numbering of its continuous dofs, then the prolongation of its two subspace matrices to the
  // finest mesh -- the chains refine[L-2] (... (refine[l] S_l)) were the bulk of the geometry build when run one after the
  // other (50 of 55 ms at L=7).  Same products in the same order: bit for bit the sequential result.
  auto& full = g.subspaces["full"];
  auto& dir = g.subspaces["dirichlet"];
  full.resize(L);
  dir.resize(L);
  auto level_job = [&](int l) {
    const std::vector<Tri>& Tl = tris[l];
    const int nel = (int)Tl.size();
    std::map<int, int> dof_of_vertex;  // vertex id (incl. midpoint ids) -> dof
    std::map<Edge, int> edge_count;
    for (const Tri& t : Tl)
      for (int i = 0; i < 3; ++i) edge_count[mk(t.v[i], t.v[(i + 1) % 3])]++;
    std::vector<char> bnd;
    auto dof = [&](int vid) {
      auto it = dof_of_vertex.find(vid);
      if (it != dof_of_vertex.end()) return it->second;
      int d = (int)dof_of_vertex.size();
      dof_of_vertex[vid] = d;
      bnd.push_back(0);
      return d;
    };
    std::vector<int> node_dof((size_t)7 * nel);
    for (int e = 0; e < nel; ++e) {
      const Tri& t = Tl[e];
      for (int i = 0; i < 3; ++i) node_dof[7 * e + i] = dof(t.v[i]);
      for (int i = 0; i < 3; ++i) {
        Edge ed = mk(t.v[i], t.v[(i + 1) % 3]);
        int d = dof(mid[l][ed]);
        node_dof[7 * e + 3 + i] = d;
        if (edge_count[ed] == 1) {
          bnd[d] = 1;
          bnd[dof(ed.first)] = 1;
          bnd[dof(ed.second)] = 1;
        }
      }
    }
    const int nve = (int)dof_of_vertex.size();
    for (int e = 0; e < nel; ++e) node_dof[7 * e + 6] = nve + e;
    bnd.resize(nve + nel, 0);
    const int m = nve + nel;
    std::vector<int> dmap(m, -1);
    int md = 0;
    for (int d = 0; d < m; ++d)
      if (!bnd[d]) dmap[d] = md++;
    std::vector<Triplet> sf, sd;
    for (int r = 0; r < 7 * nel; ++r) {
      sf.push_back({r, node_dof[r], 1.0});
      if (dmap[node_dof[r]] >= 0) sd.push_back({r, dmap[node_dof[r]], 1.0});
    }
    Csr F = from_triplets(7 * nel, m, sf), D = from_triplets(7 * nel, md, sd);
    for (int k = l; k < L - 1; ++k) {
      F = spgemm(g.refine[k], F, true);
      D = spgemm(g.refine[k], D, true);
    }
    full[l] = std::move(F);
    dir[l] = std::move(D);
  };
  run_levels(L, level_job);
  phase("subspaces");
  return g;
}


// ---------------------------------------------------------------------------------------------------------
// 3-D broken Q_k hexahedra on [-1,1]^3 (reference: fem3d, called at src/MultiGridBarrierMPI.jl:698; "Q_k",
// k = 3 default, src:682-684).  Equispaced tensor nodes (x fastest), Newton-Cotes tensor quadrature at the
// nodes, octree refinement with child index cx + 2 cy + 4 cz.  The 1-D Lagrange basis is evaluated from its
// product formula (the oracle inverts a Vandermonde matrix instead).
namespace {

struct Lag1d {
  int k;
  std::vector<double> xi;
  explicit Lag1d(int k_) : k(k_), xi(k_ + 1) {
    for (int i = 0; i <= k; ++i) xi[i] = (double)i / k;
  }
  double val(int j, double t) const {
    double v = 1.0;
    for (int m = 0; m <= k; ++m)
      if (m != j) v *= (t - xi[m]) / (xi[j] - xi[m]);
    return v;
  }
  double der(int j, double t) const {
    double s = 0.0;
    for (int m = 0; m <= k; ++m) {
      if (m == j) continue;
      double v = 1.0 / (xi[j] - xi[m]);
      for (int r = 0; r <= k; ++r)
        if (r != j && r != m) v *= (t - xi[r]) / (xi[j] - xi[r]);
      s += v;
    }
    return s;
  }
};

}  // namespace

GeometryHost fem3d_native(int L, int k) {
  if (L < 1 || L > 8) throw ArgError("fem3d: L out of range");
  if (k < 1 || k > 3) throw ArgError("fem3d: k must be 1, 2 or 3");
  const Lag1d lg(k);
  const int m1 = k + 1, nloc = m1 * m1 * m1;
  const double wq1[3][4] = {{0.5, 0.5, 0, 0}, {1.0 / 6, 4.0 / 6, 1.0 / 6, 0}, {1.0 / 8, 3.0 / 8, 3.0 / 8, 1.0 / 8}};
  const double* wq = wq1[k - 1];
  GeometryHost g;
  g.dim = 3;
  g.block = nloc;
  g.L = L;
  // element lower corners per level, in refinement order
  std::vector<std::vector<std::array<int, 3>>> elems(L);
  elems[0].push_back({0, 0, 0});
  for (int l = 1; l < L; ++l) {
    elems[l].reserve(elems[l - 1].size() * 8);
    for (const auto& e : elems[l - 1])
      for (int c = 0; c < 8; ++c) elems[l].push_back({2 * e[0] + (c & 1), 2 * e[1] + ((c >> 1) & 1), 2 * e[2] + ((c >> 2) & 1)});
  }
  const auto& E = elems[L - 1];
  const int ne = (int)E.size();
  const int ncell = 1 << (L - 1);
  const double h = 2.0 / ncell;
  g.n = ne * nloc;
  g.x.resize((size_t)g.n * 3);
  g.w.resize(g.n);
  auto lidx = [&](int i, int j, int m) { return i + m1 * (j + m1 * m); };
  // 1-D derivative matrix at the nodes (d/dx = (1/h) d/dxi)
  std::vector<double> dB((size_t)m1 * m1);
  for (int i = 0; i < m1; ++i)
    for (int a = 0; a < m1; ++a) {
      double v = lg.der(a, lg.xi[i]) / h;
      dB[(size_t)i * m1 + a] = std::fabs(v) < 1e-13 / h ? 0.0 : v;
    }
  Csr dx(g.n, g.n), dy(g.n, g.n), dz(g.n, g.n);
  for (int e = 0; e < ne; ++e)
    for (int m = 0; m < m1; ++m)
      for (int j = 0; j < m1; ++j)
        for (int i = 0; i < m1; ++i) {
          const int r = e * nloc + lidx(i, j, m);
          g.x[3 * (size_t)r] = -1.0 + h * (E[e][0] + lg.xi[i]);
          g.x[3 * (size_t)r + 1] = -1.0 + h * (E[e][1] + lg.xi[j]);
          g.x[3 * (size_t)r + 2] = -1.0 + h * (E[e][2] + lg.xi[m]);
          g.w[r] = wq[i] * wq[j] * wq[m] * h * h * h;
          for (int a = 0; a < m1; ++a) {
            if (dB[(size_t)i * m1 + a] != 0.0) {
              dx.colidx.push_back(e * nloc + lidx(a, j, m));
              dx.vals.push_back(dB[(size_t)i * m1 + a]);
            }
            if (dB[(size_t)j * m1 + a] != 0.0) {
              dy.colidx.push_back(e * nloc + lidx(i, a, m));
              dy.vals.push_back(dB[(size_t)j * m1 + a]);
            }
            if (dB[(size_t)m * m1 + a] != 0.0) {
              dz.colidx.push_back(e * nloc + lidx(i, j, a));
              dz.vals.push_back(dB[(size_t)m * m1 + a]);
            }
          }
          dx.rowptr[r + 1] = (int)dx.colidx.size();
          dy.rowptr[r + 1] = (int)dy.colidx.size();
          dz.rowptr[r + 1] = (int)dz.colidx.size();
        }
  g.operators["dx"] = std::move(dx);
  g.operators["dy"] = std::move(dy);
  g.operators["dz"] = std::move(dz);
  g.operators["id"] = identity(g.n);
  // refine block (8 nloc x nloc) and injection (nloc x 8 nloc)
  std::vector<double> P1((size_t)2 * m1 * m1);   // [c][i][a] = basis a at the i-th node of child c
  for (int c = 0; c < 2; ++c)
    for (int i = 0; i < m1; ++i)
      for (int a = 0; a < m1; ++a) P1[((size_t)c * m1 + i) * m1 + a] = lg.val(a, 0.5 * c + 0.5 * lg.xi[i]);
  std::vector<Triplet> pblk, iblk;
  for (int c = 0; c < 8; ++c)
    for (int m = 0; m < m1; ++m)
      for (int j = 0; j < m1; ++j)
        for (int i = 0; i < m1; ++i)
          for (int cc = 0; cc < m1; ++cc)
            for (int b = 0; b < m1; ++b)
              for (int a = 0; a < m1; ++a) {
                const double v = P1[(((size_t)(c & 1)) * m1 + i) * m1 + a] * P1[(((size_t)((c >> 1) & 1)) * m1 + j) * m1 + b] *
                                 P1[(((size_t)((c >> 2) & 1)) * m1 + m) * m1 + cc];
                if (std::fabs(v) >= 1e-14) pblk.push_back({c * nloc + lidx(i, j, m), lidx(a, b, cc), v});
              }
  for (int m = 0; m < m1; ++m)
    for (int j = 0; j < m1; ++j)
      for (int i = 0; i < m1; ++i) {
        const int t[3] = {2 * i, 2 * j, 2 * m};
        int c[3], q[3];
        for (int d = 0; d < 3; ++d) {
          c[d] = t[d] > k ? 1 : 0;   // ties go to child 0
          q[d] = t[d] - c[d] * k;
        }
        iblk.push_back({lidx(i, j, m), (c[0] + 2 * c[1] + 4 * c[2]) * nloc + lidx(q[0], q[1], q[2]), 1.0});
      }
  for (int l = 0; l + 1 < L; ++l) {
    const int nel = (int)elems[l].size();
    std::vector<Triplet> r, cmat;
    r.reserve(pblk.size() * nel);
    for (int e = 0; e < nel; ++e) {
      for (const auto& t : pblk) r.push_back({e * 8 * nloc + t.r, e * nloc + t.c, t.v});
      for (const auto& t : iblk) cmat.push_back({e * nloc + t.r, e * 8 * nloc + t.c, 1.0});
    }
    g.refine.push_back(from_triplets(nel * 8 * nloc, nel * nloc, std::move(r)));
    g.coarsen.push_back(from_triplets(nel * nloc, nel * 8 * nloc, std::move(cmat)));
  }
  g.refine.push_back(identity(g.n));
  g.coarsen.push_back(identity(g.n));
  auto& full = g.subspaces["full"];
  auto& dir = g.subspaces["dirichlet"];
  full.resize(L);
  dir.resize(L);
  auto level_job = [&](int l) {      // one thread per level, as in fem2d_native
    const auto& El = elems[l];
    const int nel = (int)El.size();
    const int npts = k * (1 << l) + 1;
    const long long ntot = (long long)npts * npts * npts;
    if (ntot > 2000000000LL) throw ArgError("fem3d: too many continuous dofs for Int32");
    std::vector<int> dmap((size_t)ntot, -1);
    int md = 0;
    for (int c = 0; c < npts; ++c)
      for (int b = 0; b < npts; ++b)
        for (int a = 0; a < npts; ++a)
          if (a > 0 && a < npts - 1 && b > 0 && b < npts - 1 && c > 0 && c < npts - 1)
            dmap[(size_t)a + (size_t)npts * (b + (size_t)npts * c)] = md++;
    std::vector<Triplet> sf, sd;
    for (int e = 0; e < nel; ++e)
      for (int m = 0; m < m1; ++m)
        for (int j = 0; j < m1; ++j)
          for (int i = 0; i < m1; ++i) {
            const int r = e * nloc + lidx(i, j, m);
            const size_t gid = (size_t)(k * El[e][0] + i) + (size_t)npts * ((k * El[e][1] + j) + (size_t)npts * (k * El[e][2] + m));
            sf.push_back({r, (int)gid, 1.0});
            if (dmap[gid] >= 0) sd.push_back({r, dmap[gid], 1.0});
          }
    Csr F = from_triplets(nel * nloc, (int)ntot, sf), D = from_triplets(nel * nloc, md, sd);
    for (int kk = l; kk < L - 1; ++kk) {
      F = spgemm(g.refine[kk], F, true);
      D = spgemm(g.refine[kk], D, true);
    }
    full[l] = std::move(F);
    dir[l] = std::move(D);
  };
  run_levels(L, level_job);
  return g;
}

}  // namespace mgb
