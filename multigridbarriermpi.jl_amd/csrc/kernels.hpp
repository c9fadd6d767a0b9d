// Launchers for the gfx950 kernels of the multigrid-barrier Newton path (kernels.hip).
// Everything here is fp64 values / Int32 indices (reference: T=Float64, Ti=Int32, src:260,559).
#pragma once
#include <hip/hip_runtime.h>

namespace mgb {

// Device CSR (row-block of an HPCSparseMatrix: local CSR of the owned rows, reference layout
// test/test_dump_matrices.jl:62-71).  `group` = lanes cooperating on one row (power of two <= 64).
struct DevCsr {
  int rows = 0, cols = 0, nnz = 0, group = 1;
  int* rowptr = nullptr;
  int* colidx = nullptr;
  double* vals = nullptr;
};

// Element-local view of a CSR matrix whose rows come in element blocks (rows [e * rows_per_el, (e + 1) * rows_per_el)) that
// only touch a handful of columns each -- B = D R of a broken finite-element space: the 28 rows of a triangle touch its 14
// continuous dofs.  `ecols[e * cmax + j]` lists element e's columns (padded with its first one) and `lcol[k]` replaces the
// 4-byte column index of nonzero k by a 1-byte index into that list: the kernel stages the element's x entries in LDS once
// (one coalesced gather per element instead of one scattered 8-byte gather per nonzero) and reads 9 instead of 12 bytes per
// nonzero.  rowptr / vals are the CSR's own.
struct DevElCsr {
  int rows_per_el = 0, cmax = 0, nel = 0;
  const int* ecols = nullptr;
  const unsigned char* lcol = nullptr;
  bool valid() const { return nel > 0; }
};

// One term of the barrier, acting on columns of the n x K row-major matrix Dz:
//   kind 0 -- power cone  Q = {(q,s): s >= |q|^p}  (upstream convex_Euclidian_power), F = -log(s^(2/p) - |q|^2) - mu log s
//             on columns iq[0..nq) and `is`.  If is2 >= 0 the slack is Dz[is] + Dz[is2] (feasibility phase: the extra column
//             relaxes the cone).
//   kind 1 -- half space  {y: sum_i coef[i] y[iq[i]] + off > 0}  (upstream convex_linear with one constant row: bounds and
//             constant obstacles), F = -log(sum_i coef[i] y[iq[i]] + off); its cone distance phi is that affine form.
struct ConeSpec {
  int kind = 0;
  int nq = 0;
  int iq[3] = {0, 0, 0};
  int is = 0;
  int is2 = -1;
  double a = 2.0;   // 2/p
  double mu = 1.0;
  double coef[3] = {0.0, 0.0, 0.0};
  double off = 0.0;
  __host__ __device__ int nact() const { return kind == 1 ? nq : nq + 1 + (is2 >= 0 ? 1 : 0); }
  __host__ __device__ int nY() const { return nact() * (nact() + 1) / 2; }
  // active column a (0..nact) -> row of D
  __host__ __device__ int col(int a) const { return a < nq ? iq[a] : (a == nq ? is : is2); }
};

// Barrier of an intersection of up to three such sets (upstream `intersect`): F = sum of the terms' barriers.
// Hessian slots: term 0's (a<=b) pairs, then term 1's, ...
constexpr int kMaxCones = 3;
struct BarrierParams {
  int K = 0;
  int ncones = 1;
  ConeSpec cone[kMaxCones];
  // x-dependent exponents p(x) of the power cones (upstream convex_Euclidian_power with a function p): per node
  // and term, a = 2 / p(x_q) and mu(p(x_q)) in device arrays of n x ncones doubles; null = the terms' own constants
  const double* a_node = nullptr;
  const double* mu_node = nullptr;
  // upstream convex_piecewise (a set that varies in space: at x the intersection of the pieces selected there): per node and
  // term one byte, 0 = the term is inactive at that node (no constraint, no barrier); null = every term everywhere
  const unsigned char* term_mask = nullptr;
  __host__ __device__ bool active(int ci, long long q) const { return !term_mask || term_mask[q * ncones + ci]; }
  __host__ __device__ int nY() const {
    int s = 0;
    for (int c = 0; c < ncones; ++c) s += cone[c].nY();
    return s;
  }
};

// y = (y0 ? y0 : 0) + A x      (y may alias y0)
void launch_spmv(hipStream_t st, const DevCsr& A, const double* x, const double* y0, double* y);
// the same product through the element-local view (bitwise the same y: per row the same lanes add the same terms in the
// same order; only the x entries come from LDS)
void launch_spmv_el(hipStream_t st, const DevCsr& A, const DevElCsr& E, const double* x, const double* y0, double* y);
// ---- Float32 instantiation (csrc/kernels_f32.hip): the SpMV and barrier kernels as templates over the scalar type.  `vals` are
// the matrix values converted to float; the index arrays are A's own.  f0 comes as per-row terms in double (w F, w <c, Dz>),
// summed by the caller's double reduction.  The *_tpl_f64 launches run the T = double instantiation of the same templates
// (tests: bit for bit the production kernels).
void launch_spmv_f32(hipStream_t st, const DevCsr& A, const float* vals, const float* x, const float* y0, float* y);
void launch_barrier_f0_rows_f32(hipStream_t st, int n, BarrierParams P, const float* Dz, const float* w, const float* c,
                                double* outF, double* outC);
void launch_barrier_f1_f32(hipStream_t st, int n, BarrierParams P, const float* Dz, const float* w, const float* c, float t, float* v);
void launch_barrier_f2_f32(hipStream_t st, int n, BarrierParams P, const float* Dz, const float* w, float* Y);
void launch_to_f32(hipStream_t st, long long n, const double* in, float* out);
void launch_spmv_tpl_f64(hipStream_t st, const DevCsr& A, const double* x, const double* y0, double* y);
void launch_barrier_f1_tpl_f64(hipStream_t st, int n, BarrierParams P, const double* Dz, const double* w, const double* c, double t,
                               double* v);
void launch_barrier_f2_tpl_f64(hipStream_t st, int n, BarrierParams P, const double* Dz, const double* w, double* Y);
// out = x + alpha*y
void launch_waxpby(hipStream_t st, int n, const double* x, double alpha, const double* y, double* out);
// Reductions finish inside the producing launch (csrc/kernels.hip: grid_finish): `scratch` starts with
// kReductionHeader doubles of ticket words -- ZERO before the first launch, re-armed by every launch -- followed by the
// per-block partials (2 * f0_blocks(n) doubles for the objective kernels, f0_blocks(n) for dot / sum).  Results go to `out` (device, for a
// following collective) and / or `out_host` (pinned host memory the kernel writes directly: no copy launch); either may be null.
// out2[0] = sum_q w F(Dz_q) ; out2[1] = sum_q w <c_q, Dz_q>   (+inf / NaN if any row infeasible)
constexpr int kReductionHeader = 144;      // 9 ticket counters x 128 bytes
// Completion signal of a reduction launch (nullable pair): after the results are in `out_host`, the launch bumps the device
// counter *seq_dev and stores the new value to pinned host memory *seq_host (system-scope release).  A host that polls
// *seq_host learns of the result ~1 us after the kernel, without the 15-25 us of an interrupt-driven stream synchronisation.
struct HostSignal {
  unsigned long long* seq_dev = nullptr;
  unsigned long long* seq_host = nullptr;
};
int f0_blocks(int n);
// phi_ref (nullable, n x ncones) + frac: fraction-to-the-boundary test of a line-search trial; phi_out
// (nullable, n x ncones): per-row cone distances s^(2/p) - |q|^2 of this evaluation.
void launch_barrier_f0(hipStream_t st, int n, BarrierParams P, const double* Dz, const double* w, const double* c,
                       const double* phi_ref, double frac, double* phi_out, double* scratch, double* out2,
                       double* out2_host = nullptr, HostSignal sig = HostSignal());
// the same two sums for x = s + alpha * nstep (nstep nullable: x = s) in one launch: Dz = Dz0 + B x is written on the
// way (B: the n K x N apply_D matrix of the level, rows node-major), s_out (nullable) receives x
void launch_trial_f0(hipStream_t st, const DevCsr& B, int n, BarrierParams P, const double* s, double alpha,
                     const double* nstep, double* s_out, const double* Dz0, double* Dz, const double* w, const double* c,
                     const double* phi_ref, double frac, double* phi_out, double* scratch, double* out2,
                     double* out2_host = nullptr, HostSignal sig = HostSignal());
// Up to three trial points x_a = s + alpha[a] * nstep evaluated by ONE launch of the fused objective kernel (one pass over B):
// point a leaves x_a in s_out[a] (nullable), D(z + R x_a) in dz[a], its cone distances in phi_out[a] (nullable) and its two
// sums in out_dev[2 a], out_dev[2 a + 1] (and out_host[...], either pointer nullable).  Bitwise what `na` separate
// launch_trial_f0 calls give.  scratch: kReductionHeader + 2 * na * f0_blocks(n) doubles.
struct TrialSet {
  int na = 0;
  double alpha[3] = {0.0, 0.0, 0.0};
  double* s_out[3] = {nullptr, nullptr, nullptr};
  double* dz[3] = {nullptr, nullptr, nullptr};
  double* phi_out[3] = {nullptr, nullptr, nullptr};
  double* out_dev = nullptr;
  double* out_host = nullptr;
};
void launch_trial_set(hipStream_t st, const DevCsr& B, int n, BarrierParams P, const double* s, const double* nstep,
                      const TrialSet& T, const double* Dz0, const double* w, const double* c, const double* phi_ref, double frac,
                      double* scratch, HostSignal sig = HostSignal());
// v[q,k] = w_q (dF/dDz_k + t c[q,k])
void launch_barrier_f1(hipStream_t st, int n, BarrierParams P, const double* Dz, const double* w, const double* c,
                       double t, double* v);
// Y[q, base_c + slot(a,b)] = w_q d2F/dDz_a dDz_b over cone c's active columns (a<=b),
// slot = a*nact - a(a-1)/2 + (b-a)
void launch_barrier_f2(hipStream_t st, int n, BarrierParams P, const double* Dz, const double* w, double* Y);
// map_rows of the barrier: out[q] = F(Dz_q) (+inf outside the set);  out[q, :] = vec of the K x K Hessian from the packed
// slots Y of launch_barrier_f2 (with w = 1)
void launch_barrier_rows_F(hipStream_t st, int n, BarrierParams P, const double* Dz, double* out);
void launch_expand_hessian_rows(hipStream_t st, int n, BarrierParams P, const double* Y, double* out);
// out[0] = sum x_i y_i ; scratch of kReductionHeader + f0_blocks(n) doubles
// flag_dev / flag_host (nullable pair): *flag_host = *flag_dev, then *flag_dev = 0 (a device flag set by EARLIER launches on the
// stream travels to pinned host memory with the dot product and is re-armed)
void launch_dot(hipStream_t st, int n, const double* x, const double* y, double* scratch, double* out, double* out_host = nullptr,
                int* flag_dev = nullptr, int* flag_host = nullptr, HostSignal sig = HostSignal());
// Owner-local vectors of a sharded job: `kind[i]` = 0 another rank's interior unknown, 1 this rank's, 2 top
// (replicated).  out[0] = sum over the unknowns this rank answers for -- its own, and the top ones iff count_top -- of x_i y_i;
// summed over the ranks that is the full dot product.
void launch_dot_owned(hipStream_t st, int n, const double* x, const double* y, const int* kind, int count_top, double* scratch,
                      double* out);
// buf[i] = g[top[i]], i < ntop   (this rank's partial sums at the top unknowns, to be summed over the ranks)
void launch_gather_top(hipStream_t st, int ntop, const int* top, const double* g, double* buf);
// g[top[i]] = buf[i];  out[0] = buf[ntop] + sum_i buf[i]^2   (|g|^2 from the summed interior part and the summed top entries);
// flag_dev (nullable): out[1] = (*flag_dev != 0)
void launch_scatter_top_norm(hipStream_t st, int ntop, const int* top, const double* buf, double* g, double* out);
void launch_flag_to_double(hipStream_t st, const int* flag_dev, double* out);
// out[0] = sum x_i ; scratch of kReductionHeader + f0_blocks(n) doubles
void launch_sum(hipStream_t st, int n, const double* x, double* scratch, double* out, double* out_host = nullptr);
// flag[0] = 1 if all finite else 0
void launch_all_isfinite(hipStream_t st, int n, const double* x, int* flag);
// out = x .* y
void launch_mul(hipStream_t st, int n, const double* x, const double* y, double* out);
// gather a column of a row-major n x K matrix: out[q] = M[q*K + k]
void launch_col_extract(hipStream_t st, int n, int K, int k, const double* M, double* out);

}  // namespace mgb
