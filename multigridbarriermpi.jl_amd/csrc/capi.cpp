// C ABI of libmgb_hip.so (include/mgb_hip.h): exception firewall + handle plumbing.
#include "../../include/mgb_hip.h"

#include <chrono>
#include <cmath>
#include <cstdlib>
#include <string>
#include <thread>

#include "amg.hpp"

using namespace mgb;

struct mgb_ctx_s {
  Ctx ctx;
  explicit mgb_ctx_s(int dev) : ctx(dev) {}
};
struct mgb_geo_s {
  GeometryHost g;
};
struct mgb_vec_s {
  mgb_ctx_s* ctx;
  DevBuf<double> buf;
  int n;
};
struct mgb_csr_s {
  mgb_ctx_s* ctx;
  DevCsrOwned A;
  Csr host;      // structure + values as uploaded: SparseMatrixCSC(x) gathers (src:371) and the setup-time sparse algebra
};
struct mgb_amg_s {
  mgb_ctx_s* ctx;
  std::unique_ptr<Amg> amg;
  SolveStats stats;
  bool schedule_all = false;
  bool host_solve = false;
  bool pcg = false;
  bool upstream_stop = false;
  bool exact_centering = true;
};
struct mgb_plan_s {
  LevelPlan plan;
  int n, nY;
};
struct mgb_hostchol_s {
  MfChol ch;
};

namespace {

thread_local std::string g_err;

int fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}

// exception firewall: the status code comes from the exception TYPE (errors.hpp), never from its message
template <class F>
int guard(F&& fn) {
  try {
    fn();
    return MGB_OK;
  } catch (const HipError& e) {
    return fail(MGB_E_HIP, e.what());
  } catch (const NumericError& e) {
    return fail(MGB_E_NUMERIC, e.what());
  } catch (const std::invalid_argument& e) {      // ArgError and need()
    return fail(MGB_E_ARG, e.what());
  } catch (const std::out_of_range& e) {
    return fail(MGB_E_ARG, e.what());
  } catch (const std::bad_alloc&) {
    return fail(MGB_E_INTERNAL, "out of host memory");
  } catch (const std::exception& e) {
    return fail(MGB_E_INTERNAL, e.what());
  } catch (...) {
    return fail(MGB_E_INTERNAL, "unknown exception");
  }
}

void need(bool cond, const char* msg) {
  if (!cond) throw std::invalid_argument(msg);
}

Csr make_csr(int rows, int cols, const int32_t* rowptr, const int32_t* colidx, const double* vals, const char* what) {
  need(rows >= 0 && cols >= 0 && rowptr, "CSR: null rowptr or negative size");
  Csr A(rows, cols);
  A.rowptr.assign(rowptr, rowptr + rows + 1);
  const int nnz = rowptr[rows];
  need(nnz >= 0 && (nnz == 0 || (colidx && vals)), "CSR: null colidx/vals");
  A.colidx.assign(colidx, colidx + nnz);
  A.vals.assign(vals, vals + nnz);
  check_csr(A, what);
  return A;
}

// "sub:dirichlet:2" -> parts
std::vector<std::string> split(const std::string& s) {
  std::vector<std::string> out;
  size_t p = 0;
  for (;;) {
    size_t q = s.find(':', p);
    out.push_back(s.substr(p, q == std::string::npos ? q : q - p));
    if (q == std::string::npos) break;
    p = q + 1;
  }
  return out;
}

Csr* find_matrix(GeometryHost& g, const std::string& name, bool create) {
  auto parts = split(name);
  if (parts.size() == 2 && parts[0] == "op") {
    if (create) return &g.operators[parts[1]];
    auto it = g.operators.find(parts[1]);
    return it == g.operators.end() ? nullptr : &it->second;
  }
  if (parts.size() == 3 && parts[0] == "sub") {
    int l = atoi(parts[2].c_str());
    if (l < 0 || l >= g.L) return nullptr;
    if (create) {
      auto& v = g.subspaces[parts[1]];
      if ((int)v.size() < g.L) v.resize(g.L);
      return &v[l];
    }
    auto it = g.subspaces.find(parts[1]);
    if (it == g.subspaces.end() || l >= (int)it->second.size()) return nullptr;
    return &it->second[l];
  }
  if (parts.size() == 2 && (parts[0] == "refine" || parts[0] == "coarsen")) {
    int l = atoi(parts[1].c_str());
    if (l < 0 || l >= g.L) return nullptr;
    auto& v = parts[0] == "refine" ? g.refine : g.coarsen;
    if (create && (int)v.size() < g.L) v.resize(g.L);
    if (l >= (int)v.size()) return nullptr;
    return &v[l];
  }
  return nullptr;
}

AmgSpec make_spec(int S, const char* const* sv, int K, const char* const* D) {
  need(S >= 1 && K >= 1 && sv && D, "amg: empty state_variables or D");
  AmgSpec spec;
  for (int i = 0; i < S; ++i) {
    need(sv[2 * i] && sv[2 * i + 1], "amg: null string");
    spec.state_variables.emplace_back(sv[2 * i], sv[2 * i + 1]);
  }
  for (int k = 0; k < K; ++k) {
    need(D[2 * k] && D[2 * k + 1], "amg: null string");
    spec.D.emplace_back(D[2 * k], D[2 * k + 1]);
  }
  return spec;
}

BarrierParams make_params_cones(int K, int ncones, const int* nq, const int* idx_q, const int* idx_s,
                                const int* idx_s2, const double* p) {
  need(ncones >= 1 && ncones <= 2 && nq && idx_q && idx_s && p, "barrier: 1 or 2 cones expected");
  BarrierParams P;
  P.K = K;
  P.ncones = ncones;
  for (int c = 0; c < ncones; ++c) {
    ConeSpec& S = P.cone[c];
    need(nq[c] >= 1 && nq[c] <= 3, "barrier: nq must be 1..3");
    need(p[c] >= 1.0 && std::isfinite(p[c]), "barrier: p must be >= 1");
    S.nq = nq[c];
    for (int i = 0; i < nq[c]; ++i) {
      need(idx_q[3 * c + i] >= 0 && idx_q[3 * c + i] < K, "barrier: idx_q out of range");
      S.iq[i] = idx_q[3 * c + i];
    }
    need(idx_s[c] >= 0 && idx_s[c] < K, "barrier: idx_s out of range");
    S.is = idx_s[c];
    S.is2 = idx_s2 ? idx_s2[c] : -1;
    need(S.is2 < K, "barrier: idx_s2 out of range");
    if (S.is2 < 0) S.is2 = -1;
    S.a = 2.0 / p[c];
    S.mu = (p[c] == 2.0) ? 0.0 : (p[c] < 2.0 ? 1.0 : 2.0);
  }
  return P;
}

// general menu: term c is a power cone (kind 0: nq, idx_q, idx_s, idx_s2, p as above) or a half space (kind 1: columns
// idx_q[3c .. 3c + nq), coefficients coef[3c ..], constant offset off[c]; idx_s / p ignored)
BarrierParams make_params_terms(int K, int nterms, const int* kind, const int* nq, const int* idx_q, const int* idx_s,
                                const int* idx_s2, const double* p, const double* coef, const double* off) {
  need(nterms >= 1 && nterms <= kMaxCones && kind && nq && idx_q, "barrier: 1 to 3 terms expected");
  BarrierParams P;
  P.K = K;
  P.ncones = nterms;
  for (int c = 0; c < nterms; ++c) {
    if (kind[c] == 0) {
      need(idx_s && p, "barrier: power cone needs idx_s and p");
      const int is2 = idx_s2 ? idx_s2[c] : -1;
      BarrierParams one = make_params_cones(K, 1, nq + c, idx_q + 3 * c, idx_s + c, &is2, p + c);
      P.cone[c] = one.cone[0];
    } else {
      need(kind[c] == 1, "barrier: unknown term kind");
      need(coef && off && nq[c] >= 1 && nq[c] <= 3, "barrier: half space needs 1..3 columns, coef and off");
      ConeSpec& S = P.cone[c];
      S.kind = 1;
      S.nq = nq[c];
      bool any = false;
      for (int i = 0; i < nq[c]; ++i) {
        need(idx_q[3 * c + i] >= 0 && idx_q[3 * c + i] < K, "barrier: idx_q out of range");
        for (int j = 0; j < i; ++j) need(idx_q[3 * c + j] != idx_q[3 * c + i], "barrier: repeated column in a half space");
        need(std::isfinite(coef[3 * c + i]), "barrier: coefficient not finite");
        S.iq[i] = idx_q[3 * c + i];
        S.coef[i] = coef[3 * c + i];
        any = any || coef[3 * c + i] != 0.0;
      }
      need(any && std::isfinite(off[c]), "barrier: half space needs a nonzero coefficient and a finite offset");
      S.is = S.iq[0];      // unused by kind 1; keep indices in range
      S.is2 = -1;
      S.off = off[c];
      S.a = 1.0;
      S.mu = 0.0;
    }
  }
  return P;
}

BarrierParams make_params(int K, int nq, const int* idx_q, int idx_s, double p) {
  need(nq >= 1 && nq <= 3 && idx_q, "barrier: nq must be 1..3");
  int iq3[3] = {0, 0, 0};
  for (int i = 0; i < nq; ++i) iq3[i] = idx_q[i];
  return make_params_cones(K, 1, &nq, iq3, &idx_s, nullptr, &p);
}

mgb_csr_s* new_csr(mgb_ctx_s* ctx, Csr&& A) {
  hip_check(hipSetDevice(ctx->ctx.device), "hipSetDevice");
  auto* m = new mgb_csr_s{ctx, {}, {}};
  try {
    m->A.upload(A);
    m->host = std::move(A);
  } catch (...) {
    delete m;
    throw;
  }
  return m;
}

// one scalar reduction of a device vector through `launch` (dot / sum), result on the host
template <class Launch>
double reduce_scalar(mgb_vec x, Launch&& launch) {
  if (x->n == 0) return 0.0;
  hipStream_t st = x->ctx->ctx.stream;
  const int nb = f0_blocks(x->n);
  DevBuf<double> scratch;
  scratch.alloc((size_t)kReductionHeader + nb + 1);      // ticket words | partials | result
  hip_check(hipMemsetAsync(scratch.p, 0, kReductionHeader * sizeof(double), st), "memset ticket");
  launch(st, scratch.p, scratch.p + kReductionHeader + nb);
  hip_check(hipGetLastError(), "reduction launch");
  hip_check(hipStreamSynchronize(st), "sync reduction");
  double out = 0.0;
  hip_check(hipMemcpy(&out, scratch.p + kReductionHeader + nb, sizeof(double), hipMemcpyDeviceToHost), "D2H");
  return out;
}

}  // namespace

extern "C" {

const char* mgb_last_error(void) { return g_err.c_str(); }
int mgb_version(void) { return 100; }
int mgb_device_count(void) {
  int c = 0;
  if (hipGetDeviceCount(&c) != hipSuccess) return 0;
  return c;
}

int mgb_ctx_create(int device_id, mgb_ctx* out) {
  return guard([&] {
    need(out, "null out");
    *out = new mgb_ctx_s(device_id);
  });
}
int mgb_ctx_destroy(mgb_ctx ctx) {
  return guard([&] { delete ctx; });
}
int mgb_ctx_synchronize(mgb_ctx ctx) {
  return guard([&] {
    need(ctx, "null ctx");
    hip_check(hipStreamSynchronize(ctx->ctx.stream), "sync");
  });
}

int mgb_ctx_set_comm(mgb_ctx ctx, int rank, int world, mgb_allreduce_fn fn, void* user) {
  return guard([&] {
    need(ctx && world >= 1 && rank >= 0 && rank < world && (world == 1 || fn), "ctx_set_comm: bad arguments");
    ctx->ctx.drop_comm();
    ctx->ctx.rank = rank;
    ctx->ctx.world = world;
    ctx->ctx.allreduce = fn;
    ctx->ctx.allreduce_user = user;
  });
}
int mgb_rccl_unique_id(char* out128) {
  return guard([&] {
    need(out128, "rccl_unique_id: null output");
    rccl_unique_id(out128);
  });
}
int mgb_ctx_set_comm_rccl(mgb_ctx ctx, const char* unique_id128, int rank, int world) {
  return guard([&] {
    need(ctx, "null ctx");
    ctx->ctx.set_comm_rccl(unique_id128, rank, world);
  });
}
int mgb_ctx_comm_stats(mgb_ctx ctx, long long* calls, double* bytes) {
  return guard([&] {
    need(ctx, "null ctx");
    if (calls) *calls = ctx->ctx.n_allreduce;
    if (bytes) *bytes = ctx->ctx.allreduce_bytes;
  });
}
int mgb_shard_rows(int rank, int world, int n, int block, int* r0, int* r1) {
  return guard([&] {
    need(r0 && r1, "shard_rows: null output");
    shard_rows(rank, world, n, block, r0, r1);
  });
}

int mgb_fem1d_native(int L, mgb_geo* out) {
  return guard([&] {
    need(out, "null out");
    auto* g = new mgb_geo_s;
    try {
      g->g = fem1d_native(L);
    } catch (...) {
      delete g;
      throw;
    }
    *out = g;
  });
}
int mgb_fem2d_native(int L, const double* K, int nK_rows, mgb_geo* out) {
  return guard([&] {
    need(out, "null out");
    auto* g = new mgb_geo_s;
    try {
      g->g = fem2d_native(L, K, nK_rows);
    } catch (...) {
      delete g;
      throw;
    }
    *out = g;
  });
}
int mgb_fem3d_native(int L, int k, mgb_geo* out) {
  return guard([&] {
    need(out, "null out");
    auto* g = new mgb_geo_s;
    try {
      g->g = fem3d_native(L, k);
    } catch (...) {
      delete g;
      throw;
    }
    *out = g;
  });
}
int mgb_geo_create(int n, int dim, int L, int block, const double* x, const double* w, mgb_geo* out) {
  return guard([&] {
    need(out && x && w && n > 0 && dim >= 1 && dim <= 3 && L >= 1 && block >= 1, "geo_create: bad arguments");
    auto* g = new mgb_geo_s;
    g->g.n = n;
    g->g.dim = dim;
    g->g.L = L;
    g->g.block = block;
    g->g.x.assign(x, x + (size_t)n * dim);
    g->g.w.assign(w, w + n);
    *out = g;
  });
}
int mgb_geo_set_matrix(mgb_geo g, const char* name, int rows, int cols, const int32_t* rowptr, const int32_t* colidx,
                       const double* vals) {
  return guard([&] {
    need(g && name, "null argument");
    Csr A = make_csr(rows, cols, rowptr, colidx, vals, name);
    Csr* dst = find_matrix(g->g, name, true);
    need(dst != nullptr, "geo_set_matrix: unknown matrix name");
    *dst = std::move(A);
  });
}
int mgb_geo_destroy(mgb_geo g) {
  return guard([&] { delete g; });
}
int mgb_geo_dims(mgb_geo g, int* n, int* dim, int* L, int* block) {
  return guard([&] {
    need(g, "null geo");
    if (n) *n = g->g.n;
    if (dim) *dim = g->g.dim;
    if (L) *L = g->g.L;
    if (block) *block = g->g.block;
  });
}
int mgb_geo_get_xw(mgb_geo g, double* x, double* w) {
  return guard([&] {
    need(g, "null geo");
    if (x) std::copy(g->g.x.begin(), g->g.x.end(), x);
    if (w) std::copy(g->g.w.begin(), g->g.w.end(), w);
  });
}
int mgb_geo_matrix_info(mgb_geo g, const char* name, int* rows, int* cols, int* nnz) {
  return guard([&] {
    need(g && name, "null argument");
    Csr* A = find_matrix(g->g, name, false);
    need(A != nullptr, "geo_matrix_info: unknown matrix name");
    if (rows) *rows = A->rows;
    if (cols) *cols = A->cols;
    if (nnz) *nnz = A->nnz();
  });
}
int mgb_geo_matrix_get(mgb_geo g, const char* name, int32_t* rowptr, int32_t* colidx, double* vals) {
  return guard([&] {
    need(g && name, "null argument");
    Csr* A = find_matrix(g->g, name, false);
    need(A != nullptr, "geo_matrix_get: unknown matrix name");
    if (rowptr) std::copy(A->rowptr.begin(), A->rowptr.end(), rowptr);
    if (colidx) std::copy(A->colidx.begin(), A->colidx.end(), colidx);
    if (vals) std::copy(A->vals.begin(), A->vals.end(), vals);
  });
}

// ---- device vectors / matrices
int mgb_vec_create(mgb_ctx ctx, int n, const double* host, mgb_vec* out) {
  return guard([&] {
    need(ctx && out && n >= 0, "vec_create: bad arguments");
    hip_check(hipSetDevice(ctx->ctx.device), "hipSetDevice");
    auto* v = new mgb_vec_s{ctx, {}, n};
    try {
      v->buf.alloc(n);
      if (host) v->buf.upload(host, n);
      else if (n)  // stream-ordered: a null-stream memset is not ordered against the non-blocking context stream
        hip_check(hipMemsetAsync(v->buf.p, 0, (size_t)n * sizeof(double), ctx->ctx.stream), "memset");
    } catch (...) {
      delete v;
      throw;
    }
    *out = v;
  });
}
int mgb_vec_free(mgb_vec v) {
  return guard([&] { delete v; });
}
int mgb_vec_len(mgb_vec v, int* n) {
  return guard([&] {
    need(v && n, "null argument");
    *n = v->n;
  });
}
int mgb_vec_upload(mgb_vec v, const double* host) {
  return guard([&] {
    need(v && host, "null argument");
    hip_check(hipStreamSynchronize(v->ctx->ctx.stream), "sync");
    v->buf.upload(host, v->n);
  });
}
int mgb_vec_download(mgb_vec v, double* host) {
  return guard([&] {
    need(v && host, "null argument");
    hip_check(hipStreamSynchronize(v->ctx->ctx.stream), "sync");
    v->buf.download(host, v->n);
  });
}
int mgb_csr_create(mgb_ctx ctx, int rows, int cols, const int32_t* rowptr, const int32_t* colidx, const double* vals,
                   mgb_csr* out) {
  return guard([&] {
    need(ctx && out, "null argument");
    hip_check(hipSetDevice(ctx->ctx.device), "hipSetDevice");
    Csr A = make_csr(rows, cols, rowptr, colidx, vals, "mgb_csr_create");
    *out = new_csr(ctx, std::move(A));
  });
}
int mgb_csr_free(mgb_csr A) {
  return guard([&] { delete A; });
}
int mgb_csr_dims(mgb_csr A, int* rows, int* cols, int* nnz) {
  return guard([&] {
    need(A, "null argument");
    if (rows) *rows = A->host.rows;
    if (cols) *cols = A->host.cols;
    if (nnz) *nnz = A->host.nnz();
  });
}
int mgb_csr_get(mgb_csr A, int32_t* rowptr, int32_t* colidx, double* vals) {
  return guard([&] {
    need(A, "null argument");
    if (rowptr) std::copy(A->host.rowptr.begin(), A->host.rowptr.end(), rowptr);
    if (colidx) std::copy(A->host.colidx.begin(), A->host.colidx.end(), colidx);
    if (vals) std::copy(A->host.vals.begin(), A->host.vals.end(), vals);
  });
}
int mgb_csr_spgemm(mgb_csr A, mgb_csr B, mgb_csr* out) {
  return guard([&] {
    need(A && B && out, "null argument");
    *out = new_csr(A->ctx, spgemm(A->host, B->host));
  });
}
int mgb_csr_transpose(mgb_csr A, mgb_csr* out) {
  return guard([&] {
    need(A && out, "null argument");
    *out = new_csr(A->ctx, transpose(A->host));
  });
}
int mgb_csr_add(mgb_csr A, double alpha, mgb_csr B, mgb_csr* out) {
  return guard([&] {
    need(A && B && out, "null argument");
    *out = new_csr(A->ctx, add(A->host, alpha, B->host));
  });
}
static std::vector<const Csr*> host_list(int count, const mgb_csr* mats) {
  need(count >= 1 && mats, "empty matrix list");
  std::vector<const Csr*> v;
  for (int i = 0; i < count; ++i) {
    need(mats[i] != nullptr, "null matrix in list");
    v.push_back(&mats[i]->host);
  }
  return v;
}
int mgb_csr_hcat(int count, const mgb_csr* mats, mgb_csr* out) {
  return guard([&] {
    need(out, "null argument");
    auto v = host_list(count, mats);
    *out = new_csr(mats[0]->ctx, hcat(v));
  });
}
int mgb_csr_blockdiag(int count, const mgb_csr* mats, mgb_csr* out) {
  return guard([&] {
    need(out, "null argument");
    auto v = host_list(count, mats);
    *out = new_csr(mats[0]->ctx, blockdiag(v));
  });
}
int mgb_diag(mgb_ctx ctx, mgb_vec z, int m, int n, mgb_csr* out) {
  return guard([&] {
    need(ctx && z && out && m >= 0 && n >= 0, "diag: bad arguments");
    const int d = std::min(std::min(m, n), z->n);
    std::vector<double> h(z->n);
    hip_check(hipStreamSynchronize(ctx->ctx.stream), "sync");
    z->buf.download(h.data(), z->n);
    Csr A(m, n);
    for (int i = 0; i < m; ++i) {
      if (i < d) {
        A.colidx.push_back(i);
        A.vals.push_back(h[i]);
      }
      A.rowptr[i + 1] = (int)A.colidx.size();
    }
    *out = new_csr(ctx, std::move(A));
  });
}
int mgb_spmv_add(mgb_csr A, mgb_vec x, mgb_vec y0, mgb_vec y) {
  return guard([&] {
    need(A && x && y, "null argument");
    need(A->A.view.cols == x->n && A->A.view.rows == y->n && (!y0 || y0->n == y->n), "spmv: shape mismatch");
    launch_spmv(A->ctx->ctx.stream, A->A.view, x->buf.p, y0 ? y0->buf.p : nullptr, y->buf.p);
    hip_check(hipGetLastError(), "spmv launch");
  });
}
int mgb_spmv(mgb_csr A, mgb_vec x, mgb_vec y) { return mgb_spmv_add(A, x, nullptr, y); }
int mgb_dot(mgb_vec x, mgb_vec y, double* out) {
  return guard([&] {
    need(x && y && out && x->n == y->n, "dot: shape mismatch");
    *out = reduce_scalar(x, [&](hipStream_t st, double* parts, double* res) { launch_dot(st, x->n, x->buf.p, y->buf.p, parts, res); });
  });
}
int mgb_norm(mgb_vec x, double* out) {
  return guard([&] {
    need(x && out, "null argument");
    *out = std::sqrt(reduce_scalar(x, [&](hipStream_t st, double* parts, double* res) { launch_dot(st, x->n, x->buf.p, x->buf.p, parts, res); }));
  });
}
int mgb_sum(mgb_vec x, double* out) {
  return guard([&] {
    need(x && out, "null argument");
    *out = reduce_scalar(x, [&](hipStream_t st, double* parts, double* res) { launch_sum(st, x->n, x->buf.p, parts, res); });
  });
}
int mgb_col_extract(mgb_vec M, int n, int K, int k, mgb_vec out) {
  return guard([&] {
    need(M && out && n >= 0 && K >= 1 && k >= 0 && k < K, "col_extract: bad arguments");
    need((long long)n * K == M->n && out->n == n, "col_extract: shape mismatch");
    launch_col_extract(M->ctx->ctx.stream, n, K, k, M->buf.p, out->buf.p);
    hip_check(hipGetLastError(), "col_extract launch");
  });
}
int mgb_map_rows_barrier(int which, int K, int nterms, const int* kind, const int* nq, const int* idx_q, const int* idx_s,
                         const int* idx_s2, const double* p, const double* coef, const double* off, int n, mgb_vec Dz,
                         mgb_vec out) {
  return guard([&] {
    need(Dz && out && n >= 0 && K >= 1 && K <= 8 && which >= 0 && which <= 2, "map_rows_barrier: bad arguments");
    BarrierParams P = make_params_terms(K, nterms, kind, nq, idx_q, idx_s, idx_s2, p, coef, off);
    const long long want = which == 0 ? n : (which == 1 ? (long long)n * K : (long long)n * K * K);
    need((long long)n * K == Dz->n && out->n == want, "map_rows_barrier: shape mismatch");
    hipStream_t st = Dz->ctx->ctx.stream;
    hip_check(hipSetDevice(Dz->ctx->ctx.device), "hipSetDevice");
    if (which == 0) {
      launch_barrier_rows_F(st, n, P, Dz->buf.p, out->buf.p);
    } else {
      // the fused kernels of the Newton path with unit weights: F1 = barrier_f1(w = 1, c = 0), F2 = barrier_f2(w = 1)
      std::vector<double> ones((size_t)std::max(n, 1), 1.0);
      DevBuf<double> w;
      w.upload(ones.data(), ones.size());
      if (which == 1) {
        DevBuf<double> c;
        c.alloc((size_t)std::max(n, 1) * K);
        hip_check(hipMemsetAsync(c.p, 0, c.n * sizeof(double), st), "memset");
        if (n) launch_barrier_f1(st, n, P, Dz->buf.p, w.p, c.p, 0.0, out->buf.p);
        hip_check(hipStreamSynchronize(st), "sync");      // c, w die with this scope
      } else {
        DevBuf<double> Y;
        Y.alloc((size_t)std::max(n, 1) * P.nY());
        if (n) launch_barrier_f2(st, n, P, Dz->buf.p, w.p, Y.p);
        launch_expand_hessian_rows(st, n, P, Y.p, out->buf.p);
        hip_check(hipStreamSynchronize(st), "sync");
      }
    }
    hip_check(hipGetLastError(), "map_rows_barrier launch");
  });
}
int mgb_mul(mgb_vec x, mgb_vec y, mgb_vec out) {
  return guard([&] {
    need(x && y && out && x->n == y->n && x->n == out->n, "mul: shape mismatch");
    launch_mul(x->ctx->ctx.stream, x->n, x->buf.p, y->buf.p, out->buf.p);
  });
}
int mgb_axpy(mgb_vec x, double alpha, mgb_vec y, mgb_vec out) {
  return guard([&] {
    need(x && y && out && x->n == y->n && x->n == out->n, "axpy: shape mismatch");
    launch_waxpby(x->ctx->ctx.stream, x->n, x->buf.p, alpha, y->buf.p, out->buf.p);
  });
}
int mgb_vec_allreduce_sum(mgb_vec x) {
  return guard([&] {
    need(x, "null argument");
    x->ctx->ctx.allreduce_sum(x->buf.p, x->n, /*even_single=*/true);
    hip_check(hipStreamSynchronize(x->ctx->ctx.stream), "sync allreduce");      // a library-owned communicator works on the stream
  });
}
int mgb_all_isfinite(mgb_vec x, int* out) {
  return guard([&] {
    need(x && out, "null argument");
    hipStream_t st = x->ctx->ctx.stream;
    DevBuf<int> flag;
    flag.alloc(1);
    launch_all_isfinite(st, x->n, x->buf.p, flag.p);
    hip_check(hipStreamSynchronize(st), "sync");
    int h = 0;
    flag.download(&h, 1);
    *out = h != 0;
  });
}

// ---- AMG
int mgb_amg_create(mgb_ctx ctx, mgb_geo g, int S, const char* const* state_vars, int K, const char* const* D, int nq,
                   const int* idx_q, int idx_s, double p, mgb_amg* out) {
  return guard([&] {
    need(ctx && g && out, "null argument");
    AmgSpec spec = make_spec(S, state_vars, K, D);
    BarrierParams P = make_params(K, nq, idx_q, idx_s, p);
    auto* a = new mgb_amg_s{ctx, nullptr, {}};
    try {
      const auto t0 = std::chrono::steady_clock::now();
      a->amg.reset(new Amg(ctx->ctx, g->g, spec, P));
      a->stats.t_setup = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    } catch (...) {
      delete a;
      throw;
    }
    *out = a;
  });
}
int mgb_amg_create_cones(mgb_ctx ctx, mgb_geo g, int S, const char* const* state_vars, int K, const char* const* D,
                         int ncones, const int* nq, const int* idx_q, const int* idx_s, const int* idx_s2,
                         const double* p, mgb_amg* out) {
  return guard([&] {
    need(ctx && g && out, "null argument");
    AmgSpec spec = make_spec(S, state_vars, K, D);
    BarrierParams P = make_params_cones(K, ncones, nq, idx_q, idx_s, idx_s2, p);
    auto* a = new mgb_amg_s{ctx, nullptr, {}};
    try {
      const auto t0 = std::chrono::steady_clock::now();
      a->amg.reset(new Amg(ctx->ctx, g->g, spec, P));
      a->stats.t_setup = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    } catch (...) {
      delete a;
      throw;
    }
    *out = a;
  });
}
int mgb_amg_create_terms(mgb_ctx ctx, mgb_geo g, int S, const char* const* state_vars, int K, const char* const* D, int nterms,
                         const int* kind, const int* nq, const int* idx_q, const int* idx_s, const int* idx_s2, const double* p,
                         const double* coef, const double* off, mgb_amg* out) {
  return guard([&] {
    need(ctx && g && out, "null argument");
    AmgSpec spec = make_spec(S, state_vars, K, D);
    BarrierParams P = make_params_terms(K, nterms, kind, nq, idx_q, idx_s, idx_s2, p, coef, off);
    auto* a = new mgb_amg_s{ctx, nullptr, {}};
    try {
      const auto t0 = std::chrono::steady_clock::now();
      a->amg.reset(new Amg(ctx->ctx, g->g, spec, P));
      a->stats.t_setup = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    } catch (...) {
      delete a;
      throw;
    }
    *out = a;
  });
}
int mgb_amg_destroy(mgb_amg a) {
  return guard([&] { delete a; });
}
int mgb_amg_dims(mgb_amg a, int* n, int* S, int* K, int* L, int* nY) {
  return guard([&] {
    need(a, "null amg");
    if (n) *n = a->amg->n();
    if (S) *S = a->amg->S();
    if (K) *K = a->amg->K();
    if (L) *L = a->amg->L();
    if (nY) *nY = a->amg->params().nY();
  });
}
int mgb_amg_local_rows(mgb_amg a, int* n_global, int* row0, int* n_local) {
  return guard([&] {
    need(a, "null amg");
    if (n_global) *n_global = a->amg->n_global();
    if (row0) *row0 = a->amg->row0();
    if (n_local) *n_local = a->amg->n();
  });
}
int mgb_amg_prepare(mgb_amg a, int level) {
  return guard([&] {
    need(a && level >= -1 && level < a->amg->L(), "level out of range");
    a->amg->prepare(level);
  });
}
int mgb_amg_set_exponents(mgb_amg a, int term, const double* p_nodes) {
  return guard([&] {
    need(a && p_nodes, "set_exponents: null argument");
    a->amg->set_exponents(term, p_nodes);
  });
}
int mgb_amg_set_term_mask(mgb_amg a, const unsigned char* mask) {
  return guard([&] {
    need(a && mask, "set_term_mask: null argument");
    a->amg->set_term_mask(mask);
  });
}
int mgb_amg_set_early_stop(mgb_amg a, int col) {
  return guard([&] {
    need(a && col >= -1, "set_early_stop: bad arguments");
    a->amg->set_early_stop(col);
  });
}
int mgb_amg_f0_f32(mgb_amg a, int level, const float* s, float t, double* f0) {
  return guard([&] {
    need(a && s && f0 && level >= 0 && level < a->amg->L(), "f0_f32: bad arguments");
    *f0 = a->amg->f0_f32(level, s, t);
  });
}
int mgb_amg_f1_f32(mgb_amg a, int level, const float* s, float t, float* g) {
  return guard([&] {
    need(a && s && g && level >= 0 && level < a->amg->L(), "f1_f32: bad arguments");
    a->amg->f1_f32(level, s, t, g);
  });
}
int mgb_amg_f2_f32(mgb_amg a, int level, const float* s, float t, float* lower_vals) {
  return guard([&] {
    need(a && s && lower_vals && level >= 0 && level < a->amg->L(), "f2_f32: bad arguments");
    a->amg->f2_f32(level, s, t, lower_vals);
  });
}
int mgb_amg_f1_template_f64(mgb_amg a, int level, const double* s, double t, double* g) {
  return guard([&] {
    need(a && s && g && level >= 0 && level < a->amg->L(), "f1_template_f64: bad arguments");
    a->amg->f1_tpl64(level, s, t, g);
  });
}
int mgb_amg_f2_template_f64(mgb_amg a, int level, const double* s, double t, double* lower_vals) {
  return guard([&] {
    need(a && s && lower_vals && level >= 0 && level < a->amg->L(), "f2_template_f64: bad arguments");
    a->amg->f2_tpl64(level, s, t, lower_vals);
  });
}
int mgb_amg_chol_info(mgb_amg a, int level, int* split_world, double* exchange_doubles, int* launches) {
  return guard([&] {
    need(a && level >= 0 && level < a->amg->L(), "level out of range");
    a->amg->chol_info(level, split_world, exchange_doubles, launches);
  });
}
int mgb_amg_chol_values_local(mgb_amg a, int level, int* yes) {
  return guard([&] {
    need(a && yes && level >= 0 && level < a->amg->L(), "level out of range");
    *yes = a->amg->chol_values_local(level) ? 1 : 0;
  });
}
int mgb_amg_level_size(mgb_amg a, int level, int* N, int* nnz_lower) {
  return guard([&] {
    need(a && level >= 0 && level < a->amg->L(), "level out of range");
    if (N) *N = a->amg->plan(level).N;
    if (nnz_lower) *nnz_lower = a->amg->plan(level).Apat.nnz();
  });
}
int mgb_amg_hessian_pattern(mgb_amg a, int level, int32_t* rowptr, int32_t* colidx) {
  return guard([&] {
    need(a && level >= 0 && level < a->amg->L(), "level out of range");
    const Csr& A = a->amg->plan(level).Apat;
    if (rowptr) std::copy(A.rowptr.begin(), A.rowptr.end(), rowptr);
    if (colidx) std::copy(A.colidx.begin(), A.colidx.end(), colidx);
  });
}
int mgb_amg_set_c(mgb_amg a, const double* c) {
  return guard([&] {
    need(a && c, "null argument");
    a->amg->set_c(c);
  });
}
int mgb_amg_set_z(mgb_amg a, const double* z) {
  return guard([&] {
    need(a && z, "null argument");
    a->amg->set_z(z);
  });
}
int mgb_amg_get_z(mgb_amg a, double* z) {
  return guard([&] {
    need(a && z, "null argument");
    a->amg->get_z(z);
  });
}
int mgb_amg_apply_D(mgb_amg a, int level, const double* s, double* Dz) {
  return guard([&] {
    need(a && s && Dz && level >= 0 && level < a->amg->L(), "apply_D: bad arguments");
    a->amg->apply_D(level, s, Dz);
  });
}
int mgb_amg_f0(mgb_amg a, int level, const double* s, double t, double* y, double* parts2) {
  return guard([&] {
    need(a && s && y && level >= 0 && level < a->amg->L(), "f0: bad arguments");
    *y = a->amg->f0(level, s, t, parts2);
  });
}
int mgb_amg_f0_trial(mgb_amg a, int level, const double* s_ref, const double* s, double t, double* y) {
  return guard([&] {
    need(a && s_ref && s && y && level >= 0 && level < a->amg->L(), "f0_trial: bad arguments");
    *y = a->amg->f0_trial(level, s_ref, s, t);
  });
}
int mgb_amg_f1(mgb_amg a, int level, const double* s, double t, double* g) {
  return guard([&] {
    need(a && s && g && level >= 0 && level < a->amg->L(), "f1: bad arguments");
    a->amg->f1(level, s, t, g);
  });
}
int mgb_amg_f2(mgb_amg a, int level, const double* s, double t, double* lower_vals) {
  return guard([&] {
    need(a && s && lower_vals && level >= 0 && level < a->amg->L(), "f2: bad arguments");
    a->amg->f2(level, s, t, lower_vals);
  });
}
int mgb_amg_solve_linear(mgb_amg a, int level, const double* lower_vals, const double* g, double* x) {
  return guard([&] {
    need(a && lower_vals && g && x && level >= 0 && level < a->amg->L(), "solve_linear: bad arguments");
    if (!a->amg->solve_host(level, lower_vals, g, x)) throw NumericError("MfChol: matrix is not positive definite");
  });
}
int mgb_amg_solve_linear_gpu(mgb_amg a, int level, const double* lower_vals, const double* g, double* x) {
  return guard([&] {
    need(a && lower_vals && g && x && level >= 0 && level < a->amg->L(), "solve_linear_gpu: bad arguments");
    if (!a->amg->solve_device(level, lower_vals, g, x)) throw NumericError("MfChol: matrix is not positive definite");
  });
}
int mgb_amg_set_solver(mgb_amg a, int solver) {
  return guard([&] {
    need(a && solver >= 0 && solver <= 2, "set_solver: 0 (device Cholesky), 1 (host Cholesky) or 2 (V-cycle-preconditioned CG)");
    a->host_solve = solver == 1;
    a->pcg = solver == 2;
    a->amg->set_pcg(a->pcg);
  });
}
int mgb_amg_set_pcg(mgb_amg a, double rtol, int maxit, int degree, int power_its, double lo_frac, double hi_frac, int chunk,
                    int fallback, int assembled_top, int giveup) {
  return guard([&] {
    need(a, "null amg");
    PcgOptions& o = a->amg->pcg_opt;
    if (rtol > 0) o.rtol = rtol;
    if (maxit > 0) o.maxit = maxit;
    if (degree > 0) {
      need(degree <= kChebMaxDegree, "set_pcg: smoothing degree must be 1..7");
      o.degree = degree;
    }
    if (power_its > 0) o.power_its = power_its;
    if (lo_frac > 0 && hi_frac > lo_frac) {
      o.lo_frac = lo_frac;
      o.hi_frac = hi_frac;
    }
    if (chunk > 0) o.chunk = chunk;
    if (fallback >= 0) o.fallback = fallback != 0;
    if (assembled_top >= 0) o.assembled_top = assembled_top != 0;
    if (giveup >= 0) o.giveup = giveup;
  });
}
int mgb_amg_sol_pcg(mgb_amg a, long long* counts4, double* time_s) {
  return guard([&] {
    need(a, "null amg");
    if (counts4) {
      counts4[0] = a->stats.pcg_solves;
      counts4[1] = a->stats.pcg_iters;
      counts4[2] = a->stats.pcg_fallbacks;
      counts4[3] = a->stats.pcg_gaveup_at;
    }
    if (time_s) *time_s = a->stats.time_pcg;
  });
}
int mgb_hessian_apply(mgb_amg a, int level, const double* s, const double* v, double* Hv, int matrix_free) {
  return guard([&] {
    need(a && s && v && Hv && level >= 0 && level < a->amg->L(), "hessian_apply: bad arguments");
    a->amg->hessian_apply(level, s, v, Hv, matrix_free != 0);
  });
}
int mgb_smooth(mgb_amg a, int level, const double* s, const double* b, double* x, int degree, int sweeps, double lmax,
               int matrix_free, double* lmax_used) {
  return guard([&] {
    need(a && s && b && x && sweeps >= 1 && level >= 0 && level < a->amg->L(), "smooth: bad arguments");
    const double used = a->amg->mg_smooth(level, s, b, x, degree, sweeps, lmax, matrix_free != 0);
    if (lmax_used) *lmax_used = used;
  });
}
int mgb_prolong(mgb_amg a, int level, const double* xc, double* xf) {
  return guard([&] {
    need(a && xc && xf && level >= 0 && level + 1 < a->amg->L(), "prolong: bad arguments");
    a->amg->mg_prolong(level, xc, xf);
  });
}
int mgb_restrict(mgb_amg a, int level, const double* rf, double* rc) {
  return guard([&] {
    need(a && rf && rc && level >= 0 && level + 1 < a->amg->L(), "restrict: bad arguments");
    a->amg->mg_restrict(level, rf, rc);
  });
}
int mgb_amg_prolongation(mgb_amg a, int level, int* rows, int* cols, int* nnz, int32_t* rowptr, int32_t* colidx, double* vals) {
  return guard([&] {
    need(a && level >= 0 && level + 1 < a->amg->L(), "prolongation: bad arguments");
    const Csr& P = a->amg->prolongation_host(level);
    if (rows) *rows = P.rows;
    if (cols) *cols = P.cols;
    if (nnz) *nnz = P.nnz();
    if (rowptr) std::copy(P.rowptr.begin(), P.rowptr.end(), rowptr);
    if (colidx) std::copy(P.colidx.begin(), P.colidx.end(), colidx);
    if (vals) std::copy(P.vals.begin(), P.vals.end(), vals);
  });
}
int mgb_amg_pcg_solve_linear(mgb_amg a, int level, const double* s, const double* g, double* x, int* iters, double* relres,
                             int* converged) {
  return guard([&] {
    need(a && s && g && x && level >= 0 && level < a->amg->L(), "pcg_solve_linear: bad arguments");
    const bool ok = a->amg->pcg_solve_linear(level, s, g, x, iters, relres);
    if (converged) *converged = ok ? 1 : 0;
  });
}
int mgb_amg_time_mg_kernels(mgb_amg a, int level, int reps, int nrot, double* ms6, double* bytes6, double* alg6) {
  return guard([&] {
    need(a && ms6 && bytes6 && alg6 && reps > 0 && nrot >= 1 && nrot <= 64 && level >= 0 && level < a->amg->L(),
         "time_mg_kernels: bad arguments");
    Amg::MgKernelTimes k = a->amg->time_mg_kernels(level, reps, nrot);
    std::copy(k.ms, k.ms + 6, ms6);
    std::copy(k.bytes, k.bytes + 6, bytes6);
    std::copy(k.alg, k.alg + 6, alg6);
  });
}
int mgb_amg_mg_info(mgb_amg a, int top, int* coarsest) {
  return guard([&] {
    need(a && top >= 0 && top < a->amg->L(), "mg_info: bad arguments");
    if (coarsest) *coarsest = a->amg->mg_coarsest(top);
  });
}
int mgb_amg_set_stop_rule(mgb_amg a, int upstream) {
  return guard([&] {
    need(a, "null amg");
    a->upstream_stop = upstream != 0;
  });
}
int mgb_amg_set_centering(mgb_amg a, int exact) {
  return guard([&] {
    need(a, "null amg");
    a->exact_centering = exact != 0;
  });
}
int mgb_amg_set_schedule(mgb_amg a, int all_levels) {
  return guard([&] {
    need(a, "null amg");
    a->schedule_all = all_levels != 0;
  });
}
int mgb_amg_solve(mgb_amg a, double tol, double t0, double kappa, int maxit, int max_newton, int verbose) {
  return guard([&] {
    need(a, "null amg");
    SolveOptions o;
    o.schedule_all = a->schedule_all;
    o.host_solve = a->host_solve;
    o.pcg = a->pcg;
    o.upstream_stop = a->upstream_stop;
    o.exact_centering = a->exact_centering;
    if (tol > 0) o.tol = tol;
    if (t0 > 0) o.t0 = t0;
    if (kappa > 1) o.kappa = kappa;
    if (maxit > 0) o.maxit = maxit;
    if (max_newton > 0) o.max_newton = max_newton;
    o.verbose = verbose;
    const double ts = a->stats.t_setup;
    a->amg->solve(o, a->stats);
    a->stats.t_setup = ts;
  });
}
int mgb_amg_sol_info(mgb_amg a, int* nt, double* t_elapsed, double* time_factor, long long* counts4) {
  return guard([&] {
    need(a, "null amg");
    if (nt) *nt = (int)a->stats.ts.size();
    if (t_elapsed) *t_elapsed = a->stats.t_elapsed;
    if (time_factor) *time_factor = a->stats.time_factor;
    if (counts4) {
      counts4[0] = a->stats.n_f0;
      counts4[1] = a->stats.n_f1;
      counts4[2] = a->stats.n_f2;
      counts4[3] = a->stats.n_factor;
    }
  });
}
int mgb_amg_sol_get(mgb_amg a, long long* its, double* ts, double* c_dot_Dz) {
  return guard([&] {
    need(a, "null amg");
    if (its) std::copy(a->stats.its.begin(), a->stats.its.end(), its);
    if (ts) std::copy(a->stats.ts.begin(), a->stats.ts.end(), ts);
    if (c_dot_Dz) std::copy(a->stats.c_dot_Dz.begin(), a->stats.c_dot_Dz.end(), c_dot_Dz);
  });
}
int mgb_amg_sol_kernels(mgb_amg a, double* ms11, double* bytes11, long long* launches11) {
  return guard([&] {
    need(a, "null amg");
    for (int i = 0; i < KC_COUNT; ++i) {
      if (ms11) ms11[i] = a->stats.kern_ms[i];
      if (bytes11) bytes11[i] = a->stats.kern_bytes[i];
      if (launches11) launches11[i] = a->stats.kern_launches[i];
    }
  });
}
int mgb_amg_time_kernels(mgb_amg a, int level, int reps, int nrot, double* ms8, double* bytes8) {
  return guard([&] {
    need(a && ms8 && bytes8 && reps > 0 && nrot >= 1 && nrot <= 64 && level >= 0 && level < a->amg->L(), "time_kernels: bad arguments");
    Amg::KernelTimes k = a->amg->time_kernels(level, reps, nrot);
    const double ms[8] = {k.apply_ms, k.f2_ms, k.assemble_ms, k.f1_ms, k.restrict_ms, k.f0_ms, k.trial_ms, k.apply_csr_ms};
    const double by[8] = {k.apply_bytes, k.f2_bytes, k.assemble_bytes, k.f1_bytes, k.restrict_bytes, k.f0_bytes, k.trial_bytes,
                          k.apply_el};      // last slot: 1 if apply_D (slot 0) ran through the element-local view
    std::copy(ms, ms + 8, ms8);
    std::copy(by, by + 8, bytes8);
  });
}

// ---- host-only helpers
int mgb_plan_prolongation(mgb_plan fine, mgb_plan coarse, int* rows, int* cols, int* nnz, int32_t* rowptr, int32_t* colidx,
                          double* vals) {
  return guard([&] {
    need(fine && coarse, "null plan");
    Csr P = build_prolongation(fine->plan.R, coarse->plan.R);
    if (rows) *rows = P.rows;
    if (cols) *cols = P.cols;
    if (nnz) *nnz = P.nnz();
    if (rowptr) std::copy(P.rowptr.begin(), P.rowptr.end(), rowptr);
    if (colidx) std::copy(P.colidx.begin(), P.colidx.end(), colidx);
    if (vals) std::copy(P.vals.begin(), P.vals.end(), vals);
  });
}
int mgb_plan_create(mgb_geo g, int S, const char* const* state_vars, int K, const char* const* D, int nq,
                    const int* idx_q, int idx_s, int level, mgb_plan* out) {
  return guard([&] {
    need(g && out, "null argument");
    AmgSpec spec = make_spec(S, state_vars, K, D);
    BarrierParams P = make_params(K, nq, idx_q, idx_s, 1.0);
    Csr Dstack = build_dstack(g->g, spec);
    auto* p = new mgb_plan_s;
    try {
      p->plan = build_level_plan(g->g, spec, Dstack, level, P);
      p->n = g->g.n;
      p->nY = P.nY();
    } catch (...) {
      delete p;
      throw;
    }
    *out = p;
  });
}
int mgb_reduction_scratch_doubles(int n_local, int max_level_unknowns, long long* out) {
  return guard([&] {
    need(out && n_local >= 0 && max_level_unknowns >= 0, "reduction_scratch: bad arguments");
    *out = (long long)reduction_scratch_doubles(n_local, max_level_unknowns);
  });
}
int mgb_plan_destroy(mgb_plan p) {
  return guard([&] { delete p; });
}
int mgb_plan_sizes(mgb_plan p, int* N, int* nnz_lower, int* nnz_T, int* nnz_B) {
  return guard([&] {
    need(p, "null plan");
    if (N) *N = p->plan.N;
    if (nnz_lower) *nnz_lower = p->plan.Apat.nnz();
    if (nnz_T) *nnz_T = p->plan.T.nnz();
    if (nnz_B) *nnz_B = p->plan.B.nnz();
  });
}
int mgb_plan_pattern(mgb_plan p, int32_t* rowptr, int32_t* colidx) {
  return guard([&] {
    need(p, "null plan");
    if (rowptr) std::copy(p->plan.Apat.rowptr.begin(), p->plan.Apat.rowptr.end(), rowptr);
    if (colidx) std::copy(p->plan.Apat.colidx.begin(), p->plan.Apat.colidx.end(), colidx);
  });
}
int mgb_plan_eval_host(mgb_plan p, const double* Y, double* lower_vals) {
  return guard([&] {
    need(p && Y && lower_vals, "null argument");
    // rows of T in chunks on the host threads (each value is one row's dot product: same result as the sequential loop)
    const Csr& T = p->plan.T;
    const int nthr = std::max(1, std::min(MfChol::threads(), T.rows / 4096));
    auto rows = [&](int r0, int r1) {
      for (int r = r0; r < r1; ++r) {
        double acc = 0;
        for (int k = T.rowptr[r]; k < T.rowptr[r + 1]; ++k) acc += T.vals[k] * Y[T.colidx[k]];
        lower_vals[r] = acc;
      }
    };
    std::vector<std::thread> th;
    for (int t = 1; t < nthr; ++t) th.emplace_back(rows, (int)((long long)T.rows * t / nthr), (int)((long long)T.rows * (t + 1) / nthr));
    rows(0, (int)((long long)T.rows / nthr));
    for (auto& x : th) x.join();
  });
}

int mgb_plan_shard(mgb_plan p, int S, int K, int rank, int world, int block, mgb_plan* out, int* r0, int* r1) {
  return guard([&] {
    need(p && out && r0 && r1, "plan_shard: null argument");
    shard_rows(rank, world, p->n, block, r0, r1);
    auto* q = new mgb_plan_s;
    try {
      q->plan = shard_level_plan(p->plan, p->n, S, K, p->nY, *r0, *r1);
      q->n = *r1 - *r0;
      q->nY = p->nY;
    } catch (...) {
      delete q;
      throw;
    }
    *out = q;
  });
}
int mgb_plan_apply_B_host(mgb_plan p, const double* s, double* Bs) {
  return guard([&] {
    need(p && s && Bs, "null argument");
    spmv_host(p->plan.B, s, Bs);
  });
}
int mgb_plan_apply_BT_host(mgb_plan p, const double* v, double* g) {
  return guard([&] {
    need(p && v && g, "null argument");
    spmv_host(p->plan.BT, v, g);
  });
}

int mgb_plan_chol_bench(mgb_plan p, const double* Y, int dim, int reps, double* seconds_per_factor,
                        double* seconds_per_solve, double* flops, double* front_doubles, double* residual) {
  return guard([&] {
    need(p && Y && reps > 0, "chol_bench: bad arguments");
    const LevelPlan& pl = p->plan;
    std::vector<double> vals(pl.Apat.nnz());
    spmv_host(pl.T, Y, vals.data());
    MfChol ch;
    ch.analyze(pl.Apat, pl.coords.data(), dim);
    auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < reps; ++r)
      if (!ch.factor(vals.data())) throw NumericError("MfChol: bench matrix not SPD");
    const double tf = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() / reps;
    const int N = pl.N;
    std::vector<double> xs(N), b(N, 0.0);
    for (int i = 0; i < N; ++i) xs[i] = std::sin(0.37 * i) + 0.1;
    for (int r = 0; r < N; ++r)
      for (int k = pl.Apat.rowptr[r]; k < pl.Apat.rowptr[r + 1]; ++k) {
        const int c = pl.Apat.colidx[k];
        b[r] += vals[k] * xs[c];
        if (c != r) b[c] += vals[k] * xs[r];
      }
    std::vector<double> b0 = b;
    t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < reps; ++r) {
      b = b0;
      ch.solve(b.data());
    }
    const double ts = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() / reps;
    double mx = 0, sc = 0;
    for (int i = 0; i < N; ++i) {
      mx = std::max(mx, std::fabs(b[i] - xs[i]));
      sc = std::max(sc, std::fabs(xs[i]));
    }
    if (seconds_per_factor) *seconds_per_factor = tf;
    if (seconds_per_solve) *seconds_per_solve = ts;
    if (flops) *flops = ch.factor_flops();
    if (front_doubles) *front_doubles = (double)ch.front_doubles();
    if (residual) *residual = mx / sc;
  });
}

int mgb_plan_hostchol_create(mgb_plan p, int dim, mgb_hostchol* out) {
  return guard([&] {
    need(p && out && dim >= 1 && dim <= 3, "hostchol_create: bad arguments");
    auto* c = new mgb_hostchol_s;
    try {
      c->ch.analyze(p->plan.Apat, p->plan.coords.data(), dim);
    } catch (...) {
      delete c;
      throw;
    }
    *out = c;
  });
}
int mgb_plan_hostchol_create_ranked(mgb_plan p, int dim, int K, int world, int block, mgb_hostchol* out) {
  return guard([&] {
    need(p && out && dim >= 1 && dim <= 3 && K >= 1 && world >= 1, "hostchol_create_ranked: bad arguments");
    auto* c = new mgb_hostchol_s;
    try {
      const std::vector<unsigned long long> mask = dof_rank_masks(p->plan.B, p->n, K, world, block);
      c->ch.analyze(p->plan.Apat, p->plan.coords.data(), dim, 64, mask.empty() ? nullptr : mask.data(), world);
    } catch (...) {
      delete c;
      throw;
    }
    *out = c;
  });
}
int mgb_hostchol_rank_aligned(mgb_hostchol c, int world, int* aligned, int* top_values) {
  return guard([&] {
    need(c && aligned, "hostchol_rank_aligned: null argument");
    *aligned = c->ch.rank_aligned(world) ? 1 : 0;
    if (top_values) *top_values = *aligned ? (int)c->ch.top_value_indices(c->ch.partition(world)).size() : 0;
  });
}
int mgb_hostchol_destroy(mgb_hostchol c) {
  return guard([&] { delete c; });
}
int mgb_hostchol_info(mgb_hostchol c, int* n, int* threads, double* flops) {
  return guard([&] {
    need(c, "null argument");
    if (n) *n = c->ch.size();
    if (threads) *threads = MfChol::threads();
    if (flops) *flops = c->ch.factor_flops();
  });
}
int mgb_hostchol_factor_solve(mgb_hostchol c, const double* lower_vals, const double* g, double* x) {
  return guard([&] {
    need(c && lower_vals && g && x, "hostchol_factor_solve: null argument");
    if (!c->ch.factor(lower_vals)) throw NumericError("MfChol: matrix is not positive definite");
    if (x != g) std::copy(g, g + c->ch.size(), x);
    c->ch.solve(x);
  });
}

int mgb_hostchol_partition(mgb_hostchol c, int world, int* split_world, int cap, int* nnodes, int* owner) {
  return guard([&] {
    need(c && world >= 1, "hostchol_partition: bad arguments");
    CholPartition part = c->ch.partition(world);
    if (split_world) *split_world = part.world;
    if (nnodes) *nnodes = (int)part.owner.size();
    if (owner) std::copy(part.owner.begin(), part.owner.begin() + std::min<size_t>(cap, part.owner.size()), owner);
  });
}
int mgb_hostchol_factor_solve_dist(mgb_hostchol c, int rank, int world, mgb_allreduce_fn fn, void* user, const double* lower_vals,
                                   const double* g, double* x) {
  return guard([&] {
    need(c && lower_vals && g && x && world >= 1 && rank >= 0 && rank < world && (world == 1 || fn), "hostchol_factor_solve_dist: bad arguments");
    CholPartition part = c->ch.partition(world);
    if (x != g) std::copy(g, g + c->ch.size(), x);
    auto ar = [&](double* ptr, long long count) {
      const int rc = fn(user, ptr, count);
      if (rc != 0) throw InternalError("mgb: allreduce callback failed with code " + std::to_string(rc));
    };
    if (!c->ch.factor_solve_dist(lower_vals, x, part, rank, ar)) throw NumericError("MfChol: matrix is not positive definite");
  });
}

int mgb_hostchol_factor_solve_dist_local(mgb_hostchol c, int rank, int world, mgb_allreduce_fn fn, void* user,
                                         const double* local_lower_vals, const double* g, double* x) {
  return guard([&] {
    need(c && local_lower_vals && g && x && world >= 2 && rank >= 0 && rank < world && fn, "hostchol_factor_solve_dist_local: bad arguments");
    CholPartition part = c->ch.partition(world);
    if (x != g) std::copy(g, g + c->ch.size(), x);
    auto ar = [&](double* ptr, long long count) {
      const int rc = fn(user, ptr, count);
      if (rc != 0) throw InternalError("mgb: allreduce callback failed with code " + std::to_string(rc));
    };
    if (!c->ch.factor_solve_dist(local_lower_vals, x, part, rank, ar, /*vals_local=*/true))
      throw NumericError("MfChol: matrix is not positive definite");
  });
}

int mgb_plan_chol_tree(mgb_plan p, int dim, int cap, int* nnodes, int* ns, int* nf, int* parent) {
  return guard([&] {
    need(p && nnodes, "chol_tree: bad arguments");
    MfChol ch;
    ch.analyze(p->plan.Apat, p->plan.coords.data(), dim);
    std::vector<int> a, b, c;
    ch.tree(a, b, c);
    *nnodes = (int)a.size();
    const int m = std::min<int>(cap, (int)a.size());
    if (ns) std::copy(a.begin(), a.begin() + m, ns);
    if (nf) std::copy(b.begin(), b.begin() + m, nf);
    if (parent) std::copy(c.begin(), c.begin() + m, parent);
  });
}

int mgb_chol_selftest(int nx, int ny, double* max_residual, double* flops, double* seconds) {
  return guard([&] {
    need(nx > 0 && ny > 0, "selftest: bad size");
    const int N = nx * ny;
    std::vector<Triplet> t;
    std::vector<double> coords((size_t)N * 2);
    for (int j = 0; j < ny; ++j)
      for (int i = 0; i < nx; ++i) {
        const int r = j * nx + i;
        coords[2 * r] = i;
        coords[2 * r + 1] = j;
        t.push_back({r, r, 6.0 + 1e-3 * ((r * 7919) % 13)});
        if (i > 0) t.push_back({r, r - 1, -1.0});
        if (j > 0) t.push_back({r, r - nx, -1.0});
        if (i > 0 && j > 0) t.push_back({r, r - nx - 1, -0.5});
      }
    Csr Lo = from_triplets(N, N, t);
    MfChol ch;
    ch.analyze(Lo, coords.data(), 2);
    const auto t0 = std::chrono::steady_clock::now();
    if (!ch.factor(Lo.vals.data())) throw NumericError("MfChol: selftest matrix not SPD");
    const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    std::vector<double> xs(N), b(N, 0.0);
    for (int i = 0; i < N; ++i) xs[i] = std::sin(0.37 * i) + 0.1;
    for (int r = 0; r < N; ++r)
      for (int k = Lo.rowptr[r]; k < Lo.rowptr[r + 1]; ++k) {
        const int c = Lo.colidx[k];
        b[r] += Lo.vals[k] * xs[c];
        if (c != r) b[c] += Lo.vals[k] * xs[r];
      }
    ch.solve(b.data());
    double mx = 0;
    for (int i = 0; i < N; ++i) mx = std::max(mx, std::fabs(b[i] - xs[i]));
    if (max_residual) *max_residual = mx;
    if (flops) *flops = ch.factor_flops();
    if (seconds) *seconds = sec;
  });
}

}  // extern "C"
