// gfx950 (CDNA4, wave64) kernels for the multigrid-barrier Newton path.
//
// Roofline: every kernel here is HBM/L2-bandwidth bound (AI ~0.1-0.3 flop/B, fp64); no MFMA.
//   spmv_kernel<G>   bytes = nnz*12 + (rows+1)*4 + 8*(touched x) + rows*8 (+ rows*8 if y0)
//   barrier_f0       bytes = n*(2K+1)*8                 (Dz, c, w)            -> 2 scalars
//   barrier_f1       bytes = n*(3K+1)*8                 (Dz, c, w -> v)
//   barrier_f2       bytes = n*(K+1+nY)*8               (Dz, w -> Y)
// Reference functions replaced: a3 apply_D (K SpMVs + hcat), a4/a5/a6 map_rows of the
// barrier F/F1/F2 (src/MultiGridBarrierMPI.jl:161-170), a8 amgb_all_isfinite (src:121-133),
// a9 dot/.* / column extract (test/test_column_extract.jl:50-66).
#include "kernels.hpp"

#include <algorithm>
#include <cmath>

namespace mgb {

namespace {

constexpr int kBlock = 256;
constexpr int kMaxBlocks = 2048;  // >= 8 blocks per CU on 256 CUs; grid-stride beyond that

inline int grid_for(long long work_items) {
  long long b = (work_items + kBlock - 1) / kBlock;
  if (b < 1) b = 1;
  if (b > kMaxBlocks) b = kMaxBlocks;
  return (int)b;
}

// ---------------------------------------------------------------- SpMV
// G lanes cooperate on one row: lane j reads nonzero j, j+G, ... (coalesced across the group and,
// because consecutive rows are adjacent in CSR storage, across the 64/G rows of a wave); the
// G partial sums are combined with a fixed-order shuffle tree -> bitwise reproducible.
constexpr int kSpmvU = 2;   // rows in flight per lane group (memory-level parallelism of the dependent rowptr -> nnz -> x chain)
template <int G>
__global__ __launch_bounds__(kBlock) void spmv_kernel(int rows, const int* __restrict__ rowptr,
                                                       const int* __restrict__ colidx,
                                                       const double* __restrict__ vals,
                                                       const double* __restrict__ x, const double* y0, double* y) {
  const int lane = threadIdx.x % G;
  const long long stride = (long long)gridDim.x * (kBlock / G);
  for (long long row0 = (long long)blockIdx.x * (kBlock / G) + threadIdx.x / G; row0 < rows; row0 += kSpmvU * stride) {
    int b[kSpmvU], e[kSpmvU];
    double acc[kSpmvU], base[kSpmvU];
#pragma unroll
    for (int u = 0; u < kSpmvU; ++u) {
      const long long row = row0 + u * stride;
      const bool ok = row < rows;
      b[u] = ok ? rowptr[row] : 0;
      e[u] = ok ? rowptr[row + 1] : 0;
      base[u] = (ok && y0 && lane == 0) ? y0[row] : 0.0;
      acc[u] = 0.0;
    }
    // first G nonzeros of every row: all loads of the U rows are independent and issued together
    int ci[kSpmvU];
    double va[kSpmvU];
#pragma unroll
    for (int u = 0; u < kSpmvU; ++u) {
      const int k = b[u] + lane;
      const bool in = k < e[u];
      ci[u] = in ? colidx[k] : -1;
      va[u] = in ? vals[k] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < kSpmvU; ++u) acc[u] = (ci[u] >= 0) ? va[u] * x[ci[u]] : 0.0;
    // longer rows: remaining nonzeros
#pragma unroll
    for (int u = 0; u < kSpmvU; ++u)
      for (int k = b[u] + lane + G; k < e[u]; k += G) acc[u] += vals[k] * x[colidx[k]];
#pragma unroll
    for (int u = 0; u < kSpmvU; ++u) {
      double a = acc[u];
#pragma unroll
      for (int o = G / 2; o > 0; o >>= 1) a += __shfl_down(a, o, G);
      const long long row = row0 + u * stride;
      if (lane == 0 && row < rows) y[row] = base[u] + a;
    }
  }
}

template <int G>
void spmv_launch(hipStream_t st, const DevCsr& A, const double* x, const double* y0, double* y) {
  const int grid = grid_for((long long)A.rows * G);     // small problems keep one row per lane group
  hipLaunchKernelGGL(spmv_kernel<G>, dim3(grid), dim3(kBlock), 0, st, A.rows, A.rowptr, A.colidx, A.vals, x, y0, y);
}

// ---------------------------------------------------------------- reductions
__device__ inline double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}

// sum over the block in a fixed order; result valid in thread 0
__device__ inline double block_sum(double v, double* lds) {
  v = wave_sum(v);
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) lds[wave] = v;
  __syncthreads();
  double r = 0.0;
  if (threadIdx.x == 0)
    for (int i = 0; i < kBlock / 64; ++i) r += lds[i];
  __syncthreads();
  return r;
}

__global__ __launch_bounds__(kBlock) void final_sum_kernel(int nparts, int nout, const double* __restrict__ partials,
                                                            double* out) {
  __shared__ double lds[kBlock / 64];
  for (int o = 0; o < nout; ++o) {
    double acc = 0.0;
    for (int i = threadIdx.x; i < nparts; i += kBlock) acc += partials[(size_t)i * nout + o];
    double r = block_sum(acc, lds);
    if (threadIdx.x == 0) out[o] = r;
  }
}

// ---------------------------------------------------------------- barrier
constexpr int kMaxK = 8;      // rows of D (capi.cpp rejects larger problems for the barrier kernels)

struct Cone {
  double q[3];
  double s, phi, sa;  // sa = s^a
  bool ok;
};

__device__ inline double pow_a(double s, double a) {
  if (a == 2.0) return s * s;
  if (a == 1.0) return s;
  return pow(s, a);
}

__device__ inline Cone load_cone(const ConeSpec& P, const double* dz) {
  Cone c;
  double qq = 0.0;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    c.q[i] = (i < P.nq) ? dz[P.iq[i]] : 0.0;
    qq += c.q[i] * c.q[i];
  }
  c.s = dz[P.is] + (P.is2 >= 0 ? dz[P.is2] : 0.0);
  c.ok = c.s > 0.0;
  c.sa = c.ok ? pow_a(c.s, P.a) : -1.0;
  c.phi = c.sa - qq;
  c.ok = c.ok && (c.phi > 0.0);
  return c;
}

// phi_ref (nullable, n x ncones): cone distances of every row at the current iterate; a trial row with
// phi < frac*phi_ref is treated as infeasible (fraction-to-the-boundary rule of the line search).
__global__ __launch_bounds__(kBlock) void barrier_f0_kernel(int n, BarrierParams P, const double* __restrict__ Dz,
                                                             const double* __restrict__ w,
                                                             const double* __restrict__ c,
                                                             const double* __restrict__ phi_ref, double frac,
                                                             double* __restrict__ phi_out, double* partials) {
  __shared__ double lds[kBlock / 64];
  double accF = 0.0, accL = 0.0;
  for (long long q = (long long)blockIdx.x * kBlock + threadIdx.x; q < n; q += (long long)gridDim.x * kBlock) {
    const double* dz = Dz + q * P.K;
    const double* cq = c + q * P.K;
    const double wq = w[q];
    double F = 0.0;
    for (int ci = 0; ci < P.ncones; ++ci) {
      Cone k = load_cone(P.cone[ci], dz);
      if (phi_ref && !(k.phi >= frac * phi_ref[q * P.ncones + ci])) k.ok = false;
      if (phi_out) phi_out[q * P.ncones + ci] = k.phi;
      F += k.ok ? (-log(k.phi) - P.cone[ci].mu * log(k.s)) : INFINITY;
    }
    accF += wq * F;
    double lin = 0.0;
    for (int j = 0; j < P.K; ++j) lin += cq[j] * dz[j];
    accL += wq * lin;
  }
  double rF = block_sum(accF, lds);
  double rL = block_sum(accL, lds);
  if (threadIdx.x == 0) {
    partials[2 * blockIdx.x] = rF;
    partials[2 * blockIdx.x + 1] = rL;
  }
}


// One objective evaluation of the line search in ONE launch (reference: f0 = apply_D, then map_rows of the barrier,
// then two dots -- SURVEY 8a a3/a4): x = s + alpha * nstep is formed on the fly (and written to s_out for the
// caller that accepts the trial), Dz = Dz0 + B x is computed for 64 nodes (64 K consecutive rows of B) per pass with
// the lane layout and summation order of spmv_kernel<G> -- bitwise the same Dz -- kept in LDS and written once, and
// the barrier terms of those nodes are accumulated straight from LDS.  Saves the waxpby and barrier_f0 launches and
// the re-read of Dz; bytes = spmv(B) + n (K + 2 + ncones [+ ncones]) 8.
constexpr int kTrialNodes = 64;
template <int G>
__global__ __launch_bounds__(kBlock) void trial_f0_kernel(int n, int N, BarrierParams P, const int* __restrict__ rowptr,
                                                           const int* __restrict__ colidx, const double* __restrict__ vals,
                                                           const double* __restrict__ s, double alpha,
                                                           const double* __restrict__ nstep, double* s_out,
                                                           const double* __restrict__ Dz0, double* Dz,
                                                           const double* __restrict__ w, const double* __restrict__ c,
                                                           const double* __restrict__ phi_ref, double frac,
                                                           double* __restrict__ phi_out, double* partials) {
  __shared__ double lds[kBlock / 64];
  __shared__ double dzs[kTrialNodes * kMaxK];
  constexpr int GR = kBlock / G;      // rows per pass
  const int K = P.K, lane = threadIdx.x % G, grp = threadIdx.x / G;
  if (s_out)
    for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < N; i += (long long)gridDim.x * kBlock)
      s_out[i] = s[i] + alpha * nstep[i];
  auto xval = [&](int j) { return nstep ? s[j] + alpha * nstep[j] : s[j]; };
  double accF = 0.0, accL = 0.0;
  const int nchunks = (n + kTrialNodes - 1) / kTrialNodes;
  for (int ch = blockIdx.x; ch < nchunks; ch += gridDim.x) {
    const int q0 = ch * kTrialNodes, nq = min(kTrialNodes, n - q0), nrows = nq * K;
    const long long r0 = (long long)q0 * K;
    for (int rr0 = grp; rr0 < nrows; rr0 += kSpmvU * GR) {
      int b[kSpmvU], e[kSpmvU], ci[kSpmvU];
      double acc[kSpmvU], base[kSpmvU], va[kSpmvU];
#pragma unroll
      for (int u = 0; u < kSpmvU; ++u) {
        const int rr = rr0 + u * GR;
        const bool ok = rr < nrows;
        b[u] = ok ? rowptr[r0 + rr] : 0;
        e[u] = ok ? rowptr[r0 + rr + 1] : 0;
        base[u] = (ok && lane == 0) ? Dz0[r0 + rr] : 0.0;
      }
#pragma unroll
      for (int u = 0; u < kSpmvU; ++u) {
        const int k = b[u] + lane;
        const bool in = k < e[u];
        ci[u] = in ? colidx[k] : -1;
        va[u] = in ? vals[k] : 0.0;
      }
#pragma unroll
      for (int u = 0; u < kSpmvU; ++u) acc[u] = (ci[u] >= 0) ? va[u] * xval(ci[u]) : 0.0;
#pragma unroll
      for (int u = 0; u < kSpmvU; ++u)
        for (int k = b[u] + lane + G; k < e[u]; k += G) acc[u] += vals[k] * xval(colidx[k]);
#pragma unroll
      for (int u = 0; u < kSpmvU; ++u) {
        double a = acc[u];
#pragma unroll
        for (int o = G / 2; o > 0; o >>= 1) a += __shfl_down(a, o, G);
        const int rr = rr0 + u * GR;
        if (lane == 0 && rr < nrows) {
          const double v = base[u] + a;
          Dz[r0 + rr] = v;
          dzs[rr] = v;
        }
      }
    }
    __syncthreads();
    if ((int)threadIdx.x < nq) {
      const long long q = q0 + threadIdx.x;
      const double* dz = dzs + threadIdx.x * K;
      const double* cq = c + q * K;
      const double wq = w[q];
      double F = 0.0;
      for (int ci = 0; ci < P.ncones; ++ci) {
        Cone k = load_cone(P.cone[ci], dz);
        if (phi_ref && !(k.phi >= frac * phi_ref[q * P.ncones + ci])) k.ok = false;
        if (phi_out) phi_out[q * P.ncones + ci] = k.phi;
        F += k.ok ? (-log(k.phi) - P.cone[ci].mu * log(k.s)) : INFINITY;
      }
      accF += wq * F;
      double lin = 0.0;
      for (int j = 0; j < K; ++j) lin += cq[j] * dz[j];
      accL += wq * lin;
    }
    __syncthreads();
  }
  double rF = block_sum(accF, lds);
  double rL = block_sum(accL, lds);
  if (threadIdx.x == 0) {
    partials[2 * blockIdx.x] = rF;
    partials[2 * blockIdx.x + 1] = rL;
  }
}

inline int trial_grid(int n) {
  const int b = (n + kTrialNodes - 1) / kTrialNodes;
  return b < 1 ? 1 : (b > kMaxBlocks ? kMaxBlocks : b);
}

template <int G>
void trial_launch(hipStream_t st, const DevCsr& B, int n, const BarrierParams& P, const double* s, double alpha,
                  const double* nstep, double* s_out, const double* Dz0, double* Dz, const double* w, const double* c,
                  const double* phi_ref, double frac, double* phi_out, double* partials) {
  hipLaunchKernelGGL(trial_f0_kernel<G>, dim3(trial_grid(n)), dim3(kBlock), 0, st, n, B.cols, P, B.rowptr, B.colidx, B.vals, s,
                     alpha, nstep, s_out, Dz0, Dz, w, c, phi_ref, frac, phi_out, partials);
}

// register-only helpers: the D-row indices of a cone are run-time data, so rows are picked / updated with
// unrolled selects instead of dynamically indexed local arrays (which would live in scratch) or global
// read-modify-writes (which serialise on memory latency)
__device__ inline double pick3(const double (&q)[3], int i) { return i == 0 ? q[0] : (i == 1 ? q[1] : q[2]); }

__global__ __launch_bounds__(kBlock) void barrier_f1_kernel(int n, BarrierParams P, const double* __restrict__ Dz,
                                                             const double* __restrict__ w,
                                                             const double* __restrict__ c, double t,
                                                             double* __restrict__ v) {
  for (long long q = (long long)blockIdx.x * kBlock + threadIdx.x; q < n; q += (long long)gridDim.x * kBlock) {
    const double* dz = Dz + q * P.K;
    const double* cq = c + q * P.K;
    const double wq = w[q];
    double vr[kMaxK];
#pragma unroll
    for (int j = 0; j < kMaxK; ++j) vr[j] = (j < P.K) ? wq * (t * cq[j]) : 0.0;
    for (int ci = 0; ci < P.ncones; ++ci) {
      const ConeSpec& S = P.cone[ci];
      Cone k = load_cone(S, dz);
      const double ds = S.a * pow_a(k.s, S.a - 1.0);  // d(s^a)/ds
      const double gs = wq * (-ds / k.phi - S.mu / k.s);      // same expressions as the oracle (divisions kept)
#pragma unroll
      for (int j = 0; j < kMaxK; ++j) {
        double add = 0.0;
#pragma unroll
        for (int i = 0; i < 3; ++i) add += (i < S.nq && S.iq[i] == j) ? wq * (2.0 * k.q[i] / k.phi) : 0.0;
        add += (S.is == j) ? gs : 0.0;
        add += (S.is2 == j) ? gs : 0.0;
        vr[j] += add;
      }
    }
    double* vq = v + q * P.K;
#pragma unroll
    for (int j = 0; j < kMaxK; ++j)
      if (j < P.K) vq[j] = vr[j];
  }
}

__global__ __launch_bounds__(kBlock) void barrier_f2_kernel(int n, BarrierParams P, const double* __restrict__ Dz,
                                                             const double* __restrict__ w, double* __restrict__ Y) {
  const int nY = P.nY();
  for (long long q = (long long)blockIdx.x * kBlock + threadIdx.x; q < n; q += (long long)gridDim.x * kBlock) {
    const double* dz = Dz + q * P.K;
    double* yq = Y + q * nY;
    const double wq = w[q];
    int slot = 0;
    for (int ci = 0; ci < P.ncones; ++ci) {
      const ConeSpec& S = P.cone[ci];
      Cone k = load_cone(S, dz);
      const double a = S.a;
      const double ds = a * pow_a(k.s, a - 1.0);
      const double dds = (a == 1.0) ? 0.0 : a * (a - 1.0) * pow_a(k.s, a - 2.0);
      const double ip = 1.0 / k.phi, ip2 = ip * ip;
      const double hss = -dds * ip + ds * ds * ip2 + S.mu / (k.s * k.s);
      // Hessian over (q_0..q_{nq-1}, s); a second slack column repeats the s row/column.  Upper triangle,
      // row-major, of the nact x nact block; entry (i, j) computed from its indices, no local matrix
      const int nq = S.nq, nact = S.nact();
#pragma unroll
      for (int i = 0; i < 5; ++i)
#pragma unroll
        for (int j = 0; j < 5; ++j) {
          if (i < nact && j >= i && j < nact) {
            const int ai = min(i, nq), aj = min(j, nq);
            double h;
            if (aj < nq) h = 4.0 * pick3(k.q, ai) * pick3(k.q, aj) * ip2 + (ai == aj ? 2.0 * ip : 0.0);
            else if (ai < nq) h = -2.0 * pick3(k.q, ai) * ds * ip2;
            else h = hss;
            yq[slot++] = wq * h;
          }
        }
    }
  }
}

// ---------------------------------------------------------------- vector ops
__global__ __launch_bounds__(kBlock) void waxpby_kernel(int n, const double* __restrict__ x, double alpha,
                                                         const double* __restrict__ y, double* out) {
  for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < n; i += (long long)gridDim.x * kBlock)
    out[i] = x[i] + alpha * y[i];
}

__global__ __launch_bounds__(kBlock) void mul_kernel(int n, const double* __restrict__ x, const double* __restrict__ y,
                                                      double* out) {
  for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < n; i += (long long)gridDim.x * kBlock)
    out[i] = x[i] * y[i];
}

__global__ __launch_bounds__(kBlock) void col_extract_kernel(int n, int K, int k, const double* __restrict__ M,
                                                              double* out) {
  for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < n; i += (long long)gridDim.x * kBlock)
    out[i] = M[i * K + k];
}

__global__ __launch_bounds__(kBlock) void dot_kernel(int n, const double* __restrict__ x, const double* __restrict__ y,
                                                      double* partials) {
  __shared__ double lds[kBlock / 64];
  double acc = 0.0;
  for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < n; i += (long long)gridDim.x * kBlock)
    acc += x[i] * y[i];
  double r = block_sum(acc, lds);
  if (threadIdx.x == 0) partials[blockIdx.x] = r;
}

__global__ __launch_bounds__(kBlock) void sum_kernel(int n, const double* __restrict__ x, double* partials) {
  __shared__ double lds[kBlock / 64];
  double acc = 0.0;
  for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < n; i += (long long)gridDim.x * kBlock) acc += x[i];
  double r = block_sum(acc, lds);
  if (threadIdx.x == 0) partials[blockIdx.x] = r;
}

__global__ __launch_bounds__(kBlock) void isfinite_kernel(int n, const double* __restrict__ x, int* flag) {
  int bad = 0;
  for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < n; i += (long long)gridDim.x * kBlock)
    bad |= !isfinite(x[i]);
  if (__any(bad) && (threadIdx.x & 63) == 0) atomicAnd(flag, 0);
}

}  // namespace

void launch_spmv(hipStream_t st, const DevCsr& A, const double* x, const double* y0, double* y) {
  if (A.rows == 0) return;
  switch (A.group) {
    case 1: spmv_launch<1>(st, A, x, y0, y); break;
    case 2: spmv_launch<2>(st, A, x, y0, y); break;
    case 4: spmv_launch<4>(st, A, x, y0, y); break;
    case 8: spmv_launch<8>(st, A, x, y0, y); break;
    case 16: spmv_launch<16>(st, A, x, y0, y); break;
    case 32: spmv_launch<32>(st, A, x, y0, y); break;
    default: spmv_launch<64>(st, A, x, y0, y); break;
  }
}

void launch_waxpby(hipStream_t st, int n, const double* x, double alpha, const double* y, double* out) {
  if (n == 0) return;
  hipLaunchKernelGGL(waxpby_kernel, dim3(grid_for(n)), dim3(kBlock), 0, st, n, x, alpha, y, out);
}

int f0_blocks(int n) { return std::max(grid_for(n), trial_grid(n)); }

void launch_trial_f0(hipStream_t st, const DevCsr& B, int n, BarrierParams P, const double* s, double alpha,
                     const double* nstep, double* s_out, const double* Dz0, double* Dz, const double* w, const double* c,
                     const double* phi_ref, double frac, double* phi_out, double* partials, double* out2) {
  switch (B.group) {
    case 1: trial_launch<1>(st, B, n, P, s, alpha, nstep, s_out, Dz0, Dz, w, c, phi_ref, frac, phi_out, partials); break;
    case 2: trial_launch<2>(st, B, n, P, s, alpha, nstep, s_out, Dz0, Dz, w, c, phi_ref, frac, phi_out, partials); break;
    case 4: trial_launch<4>(st, B, n, P, s, alpha, nstep, s_out, Dz0, Dz, w, c, phi_ref, frac, phi_out, partials); break;
    case 8: trial_launch<8>(st, B, n, P, s, alpha, nstep, s_out, Dz0, Dz, w, c, phi_ref, frac, phi_out, partials); break;
    case 16: trial_launch<16>(st, B, n, P, s, alpha, nstep, s_out, Dz0, Dz, w, c, phi_ref, frac, phi_out, partials); break;
    case 32: trial_launch<32>(st, B, n, P, s, alpha, nstep, s_out, Dz0, Dz, w, c, phi_ref, frac, phi_out, partials); break;
    default: trial_launch<64>(st, B, n, P, s, alpha, nstep, s_out, Dz0, Dz, w, c, phi_ref, frac, phi_out, partials); break;
  }
  hipLaunchKernelGGL(final_sum_kernel, dim3(1), dim3(kBlock), 0, st, trial_grid(n), 2, partials, out2);
}

void launch_barrier_f0(hipStream_t st, int n, BarrierParams P, const double* Dz, const double* w, const double* c,
                       const double* phi_ref, double frac, double* phi_out, double* partials, double* out2) {
  const int grid = grid_for(n);
  hipLaunchKernelGGL(barrier_f0_kernel, dim3(grid), dim3(kBlock), 0, st, n, P, Dz, w, c, phi_ref, frac, phi_out,
                     partials);
  hipLaunchKernelGGL(final_sum_kernel, dim3(1), dim3(kBlock), 0, st, grid, 2, partials, out2);
}

void launch_barrier_f1(hipStream_t st, int n, BarrierParams P, const double* Dz, const double* w, const double* c,
                       double t, double* v) {
  hipLaunchKernelGGL(barrier_f1_kernel, dim3(grid_for(n)), dim3(kBlock), 0, st, n, P, Dz, w, c, t, v);
}

void launch_barrier_f2(hipStream_t st, int n, BarrierParams P, const double* Dz, const double* w, double* Y) {
  hipLaunchKernelGGL(barrier_f2_kernel, dim3(grid_for(n)), dim3(kBlock), 0, st, n, P, Dz, w, Y);
}

void launch_dot(hipStream_t st, int n, const double* x, const double* y, double* partials, double* out) {
  const int grid = grid_for(n);
  hipLaunchKernelGGL(dot_kernel, dim3(grid), dim3(kBlock), 0, st, n, x, y, partials);
  hipLaunchKernelGGL(final_sum_kernel, dim3(1), dim3(kBlock), 0, st, grid, 1, partials, out);
}

void launch_sum(hipStream_t st, int n, const double* x, double* partials, double* out) {
  const int grid = grid_for(n);
  hipLaunchKernelGGL(sum_kernel, dim3(grid), dim3(kBlock), 0, st, n, x, partials);
  hipLaunchKernelGGL(final_sum_kernel, dim3(1), dim3(kBlock), 0, st, grid, 1, partials, out);
}

void launch_all_isfinite(hipStream_t st, int n, const double* x, int* flag) {
  (void)hipMemsetAsync(flag, 0xFF, sizeof(int), st);  // all-ones == true; kernel ANDs it to 0
  if (n == 0) return;
  hipLaunchKernelGGL(isfinite_kernel, dim3(grid_for(n)), dim3(kBlock), 0, st, n, x, flag);
}

void launch_mul(hipStream_t st, int n, const double* x, const double* y, double* out) {
  if (n == 0) return;
  hipLaunchKernelGGL(mul_kernel, dim3(grid_for(n)), dim3(kBlock), 0, st, n, x, y, out);
}

void launch_col_extract(hipStream_t st, int n, int K, int k, const double* M, double* out) {
  if (n == 0) return;
  hipLaunchKernelGGL(col_extract_kernel, dim3(grid_for(n)), dim3(kBlock), 0, st, n, K, k, M, out);
}

}  // namespace mgb
