// TEST INFRASTRUCTURE -- NOT PART OF THE PRODUCT.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
// call this; the product path (multigridbarriermpi.jl_amd/) never does.
//
// C++ / OpenMP restatement of the multigrid-barrier Newton path for the default p-Laplace problem on ALL host cores
// (SURVEY.md section 8d "CPU baseline beside it"): the statements of oracle/mgb_oracle.py (amgb_core, amgb_step on
// the finest level, newton, linesearch_backtracking, Barrier.f0 / f1 / f2 with the closed-form power-cone barrier), which
// restate what the reference reaches through MultiGridBarrier.jl (call sites src/MultiGridBarrierMPI.jl:599,666,744; Hessian
// recipe test/test_map_rows_compare.jl:102-123,165-170; solve hook test/test_instrumented_solve.jl:25-28,99).  The mesh, the
// Hessian plan (pattern + the recipe's coefficients) and the sparse Cholesky come from the host-only entry points of
// libmgb_hip.so (mgb_fem*_native, mgb_plan_*, mgb_hostchol_*: csrc/geometry.cpp, amg.cpp, mfchol.cpp) -- so this is a PORT of the
// path onto the CPU ("kind": "port"), pinned by tests/test_oracle_kats.py against the numpy oracle (z at 1e-10), not the Julia
// reference, which cannot run here.
#include <omp.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

#include "../../include/mgb_hip.h"

namespace {

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

struct Csr {
  int rows = 0, cols = 0;
  std::vector<int32_t> rp, ci;
  std::vector<double> va;
};

void check(int rc, const char* what) {
  if (rc != 0) throw std::string(what) + ": " + mgb_last_error();
}

Csr geo_matrix(mgb_geo g, const std::string& name) {
  Csr A;
  int nnz = 0;
  check(mgb_geo_matrix_info(g, name.c_str(), &A.rows, &A.cols, &nnz), name.c_str());
  A.rp.resize(A.rows + 1);
  A.ci.resize(nnz);
  A.va.resize(nnz);
  check(mgb_geo_matrix_get(g, name.c_str(), A.rp.data(), A.ci.data(), A.va.data()), name.c_str());
  return A;
}

// C = A * B restricted to what B = D R needs: row-by-row products with a dense accumulator per thread
Csr spgemm(const Csr& A, const Csr& B, int col_offset, int total_cols) {
  Csr C;
  C.rows = A.rows;
  C.cols = total_cols;
  C.rp.assign(A.rows + 1, 0);
  std::vector<std::vector<std::pair<int, double>>> rows(A.rows);
#pragma omp parallel
  {
    std::vector<double> acc(B.cols, 0.0);
    std::vector<int> mark(B.cols, -1), list;
#pragma omp for schedule(static)
    for (int r = 0; r < A.rows; ++r) {
      list.clear();
      for (int k = A.rp[r]; k < A.rp[r + 1]; ++k) {
        const int j = A.ci[k];
        for (int q = B.rp[j]; q < B.rp[j + 1]; ++q) {
          const int c = B.ci[q];
          if (mark[c] != r) {
            mark[c] = r;
            acc[c] = 0.0;
            list.push_back(c);
          }
          acc[c] += A.va[k] * B.va[q];
        }
      }
      std::sort(list.begin(), list.end());
      for (int c : list) rows[r].push_back({c + col_offset, acc[c]});
    }
  }
  for (int r = 0; r < A.rows; ++r) C.rp[r + 1] = C.rp[r] + (int)rows[r].size();
  C.ci.resize(C.rp[A.rows]);
  C.va.resize(C.rp[A.rows]);
  for (int r = 0; r < A.rows; ++r)
    for (size_t i = 0; i < rows[r].size(); ++i) {
      C.ci[C.rp[r] + i] = rows[r][i].first;
      C.va[C.rp[r] + i] = rows[r][i].second;
    }
  return C;
}

Csr transpose(const Csr& A) {
  Csr T;
  T.rows = A.cols;
  T.cols = A.rows;
  T.rp.assign(A.cols + 1, 0);
  for (int c : A.ci) T.rp[c + 1]++;
  for (int r = 0; r < T.rows; ++r) T.rp[r + 1] += T.rp[r];
  T.ci.resize(A.ci.size());
  T.va.resize(A.va.size());
  std::vector<int> pos(T.rp.begin(), T.rp.end() - 1);
  for (int r = 0; r < A.rows; ++r)
    for (int k = A.rp[r]; k < A.rp[r + 1]; ++k) {
      const int p = pos[A.ci[k]]++;
      T.ci[p] = r;
      T.va[p] = A.va[k];
    }
  return T;
}

void spmv(const Csr& A, const double* x, const double* y0, double* y) {
#pragma omp parallel for schedule(static)
  for (int r = 0; r < A.rows; ++r) {
    double s = 0;
    for (int k = A.rp[r]; k < A.rp[r + 1]; ++k) s += A.va[k] * x[A.ci[k]];
    y[r] = (y0 ? y0[r] : 0.0) + s;
  }
}

// p-Laplace power cone on rows (1 .. dim, K - 1) of Dz: F = -log(s^a - |q|^2) - mu log s, a = 2 / p (oracle PowerConeBarrier)
struct Problem {
  int n = 0, dim = 0, K = 0, N = 0, nY = 0;
  double a = 2, mu = 1;
  std::vector<double> w, c, z, Dz0;      // c: n x K; z: S n ([u; s]); Dz0: n x K
  Csr B, BT, R;
  mgb_plan plan = nullptr;
  mgb_hostchol chol = nullptr;
  int nnzA = 0;
  double w_min = 0;
};

inline double pow_a(double s, double a) { return a == 2.0 ? s * s : (a == 1.0 ? s : std::pow(s, a)); }

// objective parts at Dz: (sum w F, sum w <c, Dz>); +inf unless every row is inside the cone and (with phi_ref) keeps >= frac of
// its previous distance; phi_out (nullable) receives the distances
void f0(const Problem& P, const double* Dz, const double* phi_ref, double frac, double* phi_out, double* sumF, double* sumL) {
  const int n = P.n, K = P.K, dim = P.dim;
  double aF = 0, aL = 0;
  int bad = 0;
#pragma omp parallel for schedule(static) reduction(+ : aF, aL, bad)
  for (int q = 0; q < n; ++q) {
    const double* d = Dz + (size_t)q * K;
    double qq = 0;
    for (int i = 1; i <= dim; ++i) qq += d[i] * d[i];
    const double s = d[K - 1];
    const double phi = (s > 0 ? pow_a(s, P.a) : -1.0) - qq;
    if (phi_out) phi_out[q] = phi;
    if (!(s > 0) || !(phi > 0) || (phi_ref && !(phi >= frac * phi_ref[q]))) {
      bad++;
      continue;
    }
    aF += P.w[q] * (-std::log(phi) - P.mu * std::log(s));
    double lin = 0;
    for (int k = 0; k < K; ++k) lin += P.c[(size_t)q * K + k] * d[k];
    aL += P.w[q] * lin;
  }
  *sumF = bad ? INFINITY : aF;
  *sumL = aL;
}

// v = w (F1 + t c) (n x K), g = BT v
void f1(const Problem& P, const double* Dz, double t, std::vector<double>& v, double* g) {
  const int n = P.n, K = P.K, dim = P.dim;
#pragma omp parallel for schedule(static)
  for (int q = 0; q < n; ++q) {
    const double* d = Dz + (size_t)q * K;
    double qq = 0;
    for (int i = 1; i <= dim; ++i) qq += d[i] * d[i];
    const double s = d[K - 1], phi = pow_a(s, P.a) - qq, ds = P.a * pow_a(s, P.a - 1.0);
    double* vq = v.data() + (size_t)q * K;
    for (int k = 0; k < K; ++k) vq[k] = P.w[q] * (t * P.c[(size_t)q * K + k]);
    for (int i = 1; i <= dim; ++i) vq[i] += P.w[q] * (2.0 * d[i] / phi);
    vq[K - 1] += P.w[q] * (-ds / phi - P.mu / s);
  }
  spmv(P.BT, v.data(), nullptr, g);
}

// Y = w F2 in the plan's slot order (upper triangle, row-major, of the (q_1..q_dim, s) block)
void f2(const Problem& P, const double* Dz, std::vector<double>& Y) {
  const int n = P.n, K = P.K, dim = P.dim, nY = P.nY;
#pragma omp parallel for schedule(static)
  for (int q = 0; q < n; ++q) {
    const double* d = Dz + (size_t)q * K;
    double qq = 0;
    for (int i = 1; i <= dim; ++i) qq += d[i] * d[i];
    const double s = d[K - 1], a = P.a, phi = pow_a(s, a) - qq;
    const double ds = a * pow_a(s, a - 1.0), dds = (a == 1.0) ? 0.0 : a * (a - 1.0) * pow_a(s, a - 2.0);
    const double ip = 1.0 / phi, ip2 = ip * ip;
    const double hss = -dds * ip + ds * ds * ip2 + P.mu / (s * s);
    double* y = Y.data() + (size_t)q * nY;
    int slot = 0;
    for (int i = 0; i <= dim; ++i)
      for (int j = i; j <= dim; ++j) {
        double h;
        if (j < dim) h = 4.0 * d[1 + i] * d[1 + j] * ip2 + (i == j ? 2.0 * ip : 0.0);
        else if (i < dim) h = -2.0 * d[1 + i] * ds * ip2;
        else h = hss;
        y[slot++] = P.w[q] * h;
      }
  }
}

const double kBeta = 0.5, kArmijo = 0.1, kMinStep = 1e-8, kFrac = 0.1, kKappaGrow = 0.25;
const int kAttempts = 8, kMaxNewton = 48;

struct Work {
  std::vector<double> s, s_trial, g, g_trial, nstep, Dz, DzT, phi, phiT, v, Y, avals;
};

// Newton on the finest level at barrier parameter t (oracle newton + linesearch_backtracking, stopping_exact(0.1))
bool newton(Problem& P, Work& W, double t, long long* steps, double deadline) {
  const int N = P.N, n = P.n, K = P.K;
  std::fill(W.s.begin(), W.s.end(), 0.0);
  spmv(P.B, W.s.data(), P.Dz0.data(), W.Dz.data());
  double sF, sL;
  f0(P, W.Dz.data(), nullptr, 0, W.phi.data(), &sF, &sL);
  double y = sF + t * sL;
  if (!std::isfinite(y)) throw std::string("cpu newton: infeasible start");
  f1(P, W.Dz.data(), t, W.v, W.g.data());
  auto norm = [&](const std::vector<double>& x) {
    double a = 0;
#pragma omp parallel for reduction(+ : a)
    for (int i = 0; i < N; ++i) a += x[i] * x[i];
    return std::sqrt(a);
  };
  double gnorm = norm(W.g), ymin = y, gmin = gnorm;
  bool converged = false;
  int k = 0;
  while (k < kMaxNewton && !converged) {
    ++k;
    ++*steps;
    f2(P, W.Dz.data(), W.Y);
    check(mgb_plan_eval_host(P.plan, W.Y.data(), W.avals.data()), "plan_eval");
    if (mgb_hostchol_factor_solve(P.chol, W.avals.data(), W.g.data(), W.nstep.data()) != 0) break;
    double inc = 0;
#pragma omp parallel for reduction(+ : inc)
    for (int i = 0; i < N; ++i) inc += W.g[i] * W.nstep[i];
    if (!std::isfinite(inc)) break;
    if (inc <= 0) {
      converged = true;
      break;
    }
    auto trial = [&](double step, std::vector<double>& st, std::vector<double>& dz, std::vector<double>& ph) {
#pragma omp parallel for schedule(static)
      for (int i = 0; i < N; ++i) st[i] = W.s[i] - step * W.nstep[i];
      spmv(P.B, st.data(), P.Dz0.data(), dz.data());
      double a, b;
      f0(P, dz.data(), W.phi.data(), kFrac, ph.data(), &a, &b);
      return a + t * b;
    };
    double step = 1.0, ynext = y, gnext = gnorm;
    bool accepted = false;
    std::vector<double> s2(N), dz2((size_t)n * K), ph2(n);
    while (step >= kMinStep) {
      double yA = trial(step, W.s_trial, W.DzT, W.phiT);
      if (std::isfinite(yA) && yA <= y - kArmijo * step * inc) {
        while (step * kBeta >= kMinStep) {
          const double yB = trial(step * kBeta, s2, dz2, ph2);
          if (!(std::isfinite(yB) && yB < yA)) break;
          W.s_trial.swap(s2);
          W.DzT.swap(dz2);
          W.phiT.swap(ph2);
          yA = yB;
          step *= kBeta;
        }
        f1(P, W.DzT.data(), t, W.v, W.g_trial.data());
        const double gn = norm(W.g_trial);
        if (std::isfinite(gn)) {
          ynext = yA;
          gnext = gn;
          accepted = true;
          break;
        }
      }
      step *= kBeta;
    }
    if (accepted) {
      W.s.swap(W.s_trial);
      W.Dz.swap(W.DzT);
      W.phi.swap(W.phiT);
      W.g.swap(W.g_trial);
    }
    if (ynext >= ymin && gnext >= 0.1 * gmin) converged = true;
    y = ynext;
    gnorm = gnext;
    ymin = std::min(ymin, y);
    gmin = std::min(gmin, gnorm);
    if (deadline > 0 && now_s() > deadline + 120) break;      // hard stop far beyond the budget
  }
  // z += R s; the accepted Dz becomes Dz0 (oracle amgb_step)
  spmv(P.R, W.s.data(), P.z.data(), P.z.data());
  P.Dz0 = W.Dz;
  return converged;
}

}  // namespace

extern "C" {

// kind = 1, 2, 3 (fem1d / fem2d / fem3d with Q_k, k3 = k); default problem of the reference (D, f, g of src:735-739 and their 1-D /
// 2-D analogues), exponent p.  Runs the main phase until t_stop, or -- budget_s > 0 -- stops after the first centering that ends
// beyond the budget.  z_out (nullable): S n values [u; s].  Returns 0, or -1 with the message on stderr.
int mgb_cpu_solve(int kind, int L, int k3, double p, double budget_s, int nthreads, double* z_out, long long* newton_steps,
                  double* seconds, double* t_reached, int* threads_used) {
  mgb_geo geo = nullptr;
  Problem P;
  try {
    if (nthreads > 0) omp_set_num_threads(nthreads);
    const int dim = kind;
    if (kind == 1) check(mgb_fem1d_native(L, &geo), "fem1d");
    else if (kind == 2) check(mgb_fem2d_native(L, nullptr, 0, &geo), "fem2d");
    else if (kind == 3) check(mgb_fem3d_native(L, k3, &geo), "fem3d");
    else throw std::string("kind must be 1, 2 or 3");
    int n, d2, Lv, block;
    check(mgb_geo_dims(geo, &n, &d2, &Lv, &block), "dims");
    const int K = dim + 2;
    P.n = n;
    P.dim = dim;
    P.K = K;
    P.nY = (dim + 1) * (dim + 2) / 2;
    P.a = 2.0 / p;
    P.mu = p == 2.0 ? 0.0 : (p < 2.0 ? 1.0 : 2.0);
    std::vector<double> x((size_t)n * dim);
    P.w.resize(n);
    check(mgb_geo_get_xw(geo, x.data(), P.w.data()), "xw");
    P.w_min = *std::min_element(P.w.begin(), P.w.end());
    const char* opn[4] = {"op:id", "op:dx", "op:dy", "op:dz"};
    Csr Rd = geo_matrix(geo, "sub:dirichlet:" + std::to_string(Lv - 1)), Rf = geo_matrix(geo, "sub:full:" + std::to_string(Lv - 1));
    const int Nd = Rd.cols, Nf = Rf.cols;
    P.N = Nd + Nf;
    // R = blockdiag(Rd, Rf); B rows q K + k = (op_k R_sv)[q, :]
    P.R.rows = 2 * n;
    P.R.cols = P.N;
    P.R.rp.assign(2 * n + 1, 0);
    for (int r = 0; r < n; ++r) P.R.rp[r + 1] = Rd.rp[r + 1];
    for (int r = 0; r < n; ++r) P.R.rp[n + r + 1] = Rd.rp[n] + Rf.rp[r + 1];
    P.R.ci = Rd.ci;
    P.R.va = Rd.va;
    for (size_t i = 0; i < Rf.ci.size(); ++i) {
      P.R.ci.push_back(Rf.ci[i] + Nd);
      P.R.va.push_back(Rf.va[i]);
    }
    std::vector<Csr> rows(K);
    for (int k = 0; k < K; ++k) {
      const bool slack = k == K - 1;
      Csr op = geo_matrix(geo, slack ? opn[0] : opn[k]);
      rows[k] = spgemm(op, slack ? Rf : Rd, slack ? Nd : 0, P.N);
    }
    P.B.rows = n * K;
    P.B.cols = P.N;
    P.B.rp.assign((size_t)n * K + 1, 0);
    for (int q = 0; q < n; ++q)
      for (int k = 0; k < K; ++k) {
        const Csr& A = rows[k];
        for (int e = A.rp[q]; e < A.rp[q + 1]; ++e) {
          P.B.ci.push_back(A.ci[e]);
          P.B.va.push_back(A.va[e]);
        }
        P.B.rp[(size_t)q * K + k + 1] = (int)P.B.ci.size();
      }
    P.BT = transpose(P.B);
    // Hessian plan + Cholesky of the product's host code, on the same dof numbering
    const char* sv[4] = {"u", "dirichlet", "s", "full"};
    std::vector<const char*> D;
    const char* on[4] = {"id", "dx", "dy", "dz"};
    for (int k = 0; k < K - 1; ++k) {
      D.push_back("u");
      D.push_back(on[k]);
    }
    D.push_back("s");
    D.push_back("id");
    int iq[3] = {1, 2, 3};
    check(mgb_plan_create(geo, 2, sv, K, D.data(), dim, iq, K - 1, Lv - 1, &P.plan), "plan");
    int N2, nB;
    check(mgb_plan_sizes(P.plan, &N2, &P.nnzA, nullptr, &nB), "plan sizes");
    if (N2 != P.N || nB != (int)P.B.ci.size()) throw std::string("cpu port: plan and B disagree");
    check(mgb_plan_hostchol_create(P.plan, dim, &P.chol), "hostchol");
    // problem data: f = (0.5, 0.., 1), g = (|x|^2 | x_1, 100 | 2)  (src:737-738 and the 1-D / 2-D analogues of the oracle)
    P.c.assign((size_t)n * K, 0.0);
    P.z.assign((size_t)2 * n, 0.0);
    for (int q = 0; q < n; ++q) {
      P.c[(size_t)q * K] = 0.5;
      P.c[(size_t)q * K + K - 1] = 1.0;
      double r2 = 0;
      for (int i = 0; i < dim; ++i) r2 += x[(size_t)q * dim + i] * x[(size_t)q * dim + i];
      P.z[q] = dim == 1 ? x[q] : r2;
      P.z[n + q] = dim == 1 ? 2.0 : 100.0;
    }
    // Dz0 = Dstack z through B's building blocks: D z = sum over state vars ... evaluated as op_k z_sv directly
    P.Dz0.assign((size_t)n * K, 0.0);
    for (int k = 0; k < K; ++k) {
      const bool slack = k == K - 1;
      Csr op = geo_matrix(geo, slack ? opn[0] : opn[k]);
      std::vector<double> col(n);
      spmv(op, P.z.data() + (slack ? n : 0), nullptr, col.data());
      for (int q = 0; q < n; ++q) P.Dz0[(size_t)q * K + k] = col[q];
    }
    Work W;
    W.s.assign(P.N, 0);
    W.s_trial.assign(P.N, 0);
    W.g.assign(P.N, 0);
    W.g_trial.assign(P.N, 0);
    W.nstep.assign(P.N, 0);
    W.Dz.assign((size_t)n * K, 0);
    W.DzT.assign((size_t)n * K, 0);
    W.phi.assign(n, 0);
    W.phiT.assign(n, 0);
    W.v.assign((size_t)n * K, 0);
    W.Y.assign((size_t)n * P.nY, 0);
    W.avals.assign(P.nnzA, 0);
    // amgb_core: t0 = 0.1, kappa = 10 (sqrt on failure, squared back after an easy centering), fixed t_stop
    const double tol = std::sqrt(2.220446049250313e-16), t_begin = now_s();
    const double deadline = budget_s > 0 ? t_begin + budget_s : 0;
    double t = 0.1, kappa = 10.0;
    const double kappa0 = 10.0;
    long long steps = 0;
    bool ok0 = false;
    for (int a = 0; a < kAttempts && !ok0; ++a) ok0 = newton(P, W, t, &steps, deadline);
    if (!ok0) throw std::string("cpu port: initial centering failed");
    double t_stop = t;
    while (t_stop <= 1 / tol) t_stop *= kappa0;
    bool out_of_budget = deadline > 0 && now_s() > deadline;
    while (t < t_stop && kappa > 1 && !out_of_budget) {
      while (kappa > 1) {
        const double t1 = std::min(kappa * t, t_stop);
        std::vector<double> zs = P.z, ds = P.Dz0;
        const long long before = steps;
        const bool ok = newton(P, W, t1, &steps, deadline);
        if (ok) {
          if (steps - before <= kMaxNewton * kKappaGrow) kappa = std::min(kappa0, kappa * kappa);
          t = t1;
          break;
        }
        P.z = zs;
        P.Dz0 = ds;
        kappa = std::sqrt(kappa);
        if (kappa < 1 + 1e-3) kappa = 1.0;
      }
      out_of_budget = deadline > 0 && now_s() > deadline;
    }
    if (t < t_stop && !out_of_budget) throw std::string("cpu port: convergence failure (kappa collapsed)");
    if (z_out) std::copy(P.z.begin(), P.z.end(), z_out);
    if (newton_steps) *newton_steps = steps;
    if (seconds) *seconds = now_s() - t_begin;
    if (t_reached) *t_reached = t;
    if (threads_used) *threads_used = omp_get_max_threads();
  } catch (const std::string& e) {
    std::fprintf(stderr, "mgb_cpu_solve: %s\n", e.c_str());
    if (P.chol) mgb_hostchol_destroy(P.chol);
    if (P.plan) mgb_plan_destroy(P.plan);
    if (geo) mgb_geo_destroy(geo);
    return -1;
  }
  mgb_hostchol_destroy(P.chol);
  mgb_plan_destroy(P.plan);
  mgb_geo_destroy(geo);
  return 0;
}

}  // extern "C"
