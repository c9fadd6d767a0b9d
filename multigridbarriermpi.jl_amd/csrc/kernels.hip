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
#include "kernels_tpl.hpp"

#include <algorithm>
#include <cmath>

namespace mgb {

namespace {


// ---------------------------------------------------------------- SpMV
// y = (y0 ? y0 : 0) + A x: the double instantiation of spmv_kernel_t (kernels_tpl.hpp)
template <int G>
void spmv_launch(hipStream_t st, const DevCsr& A, const double* x, const double* y0, double* y) {
  const int grid = grid_for((long long)A.rows * G);     // small problems keep one row per lane group
  hipLaunchKernelGGL((spmv_kernel_t<G, double>), dim3(grid), dim3(kBlock), 0, st, A.rows, A.rowptr, A.colidx, A.vals, x, y0, y);
}

// Element-local SpMV (DevElCsr): a workgroup takes kElPerBlock consecutive elements per pass, stages their x entries in LDS
// (thread = (element, column slot): consecutive threads read consecutive ecols entries -- coalesced -- and gather x once per
// element column), then every lane group forms its row exactly as spmv_kernel<G> does: lane j adds nonzeros j, j + G, ...,
// fixed shuffle tree -- bitwise the same y.  bytes = nnz * 9 + (rows + 1) * 4 + nel * cmax * 12 + rows * 16.
template <int G>
__global__ __launch_bounds__(kBlock) void spmv_el_kernel(int rows, DevElCsr E, const int* __restrict__ rowptr,
                                                          const double* __restrict__ vals, const double* __restrict__ x,
                                                          const double* y0, double* y) {
  extern __shared__ double xs[];      // els_per_pass x cmax
  const int lane = threadIdx.x % G, grp = threadIdx.x / G;
  constexpr int GR = kBlock / G;
  const int rpe = E.rows_per_el;
  const int els = max(1, min(E.nel, (kBlock * 4) / rpe));      // elements per pass: ~1024 rows
  const int npass = (E.nel + els - 1) / els;
  for (int ps = xcd_block(blockIdx.x, gridDim.x); ps < npass; ps += gridDim.x) {
    const int e0 = ps * els, ne = min(els, E.nel - e0);
    __syncthreads();      // the previous pass is done with xs
    for (int idx = threadIdx.x; idx < ne * E.cmax; idx += kBlock) xs[idx] = x[E.ecols[(long long)e0 * E.cmax + idx]];
    __syncthreads();
    const long long r0 = (long long)e0 * rpe;
    const int nrows = (int)min((long long)ne * rpe, (long long)rows - r0);
    for (int rr0 = grp; rr0 < nrows; rr0 += kSpmvU * GR) {
      int b[kSpmvU], e[kSpmvU];
      double acc[kSpmvU], base[kSpmvU];
#pragma unroll
      for (int u = 0; u < kSpmvU; ++u) {
        const int rr = rr0 + u * GR;
        const bool ok = rr < nrows;
        b[u] = ok ? rowptr[r0 + rr] : 0;
        e[u] = ok ? rowptr[r0 + rr + 1] : 0;
        base[u] = (ok && y0 && lane == 0) ? y0[r0 + rr] : 0.0;
      }
#pragma unroll
      for (int u = 0; u < kSpmvU; ++u) {
        const int rr = rr0 + u * GR;
        const double* xe = xs + (rr / rpe) * E.cmax;
        const int k = b[u] + lane;
        acc[u] = (k < e[u]) ? vals[k] * xe[E.lcol[k]] : 0.0;
        for (int kk = k + G; kk < e[u]; kk += G) acc[u] += vals[kk] * xe[E.lcol[kk]];
      }
#pragma unroll
      for (int u = 0; u < kSpmvU; ++u) {
        double a = acc[u];
#pragma unroll
        for (int o = G / 2; o > 0; o >>= 1) a += __shfl_down(a, o, G);
        const int rr = rr0 + u * GR;
        if (lane == 0 && rr < nrows) y[r0 + rr] = base[u] + a;
      }
    }
  }
}

template <int G>
void spmv_el_launch(hipStream_t st, const DevCsr& A, const DevElCsr& E, const double* x, const double* y0, double* y) {
  const int els = std::max(1, std::min(E.nel, (kBlock * 4) / E.rows_per_el));
  const int npass = (E.nel + els - 1) / els;
  const int grid = std::max(1, std::min(npass, kMaxBlocks * 4));
  hipLaunchKernelGGL(spmv_el_kernel<G>, dim3(grid), dim3(kBlock), (size_t)els * E.cmax * sizeof(double), st, A.rows, E, A.rowptr,
                     A.vals, x, y0, y);
}


// ---------------------------------------------------------------- barrier
using Cone = ConeT<double>;      // kernels_tpl.hpp: load_cone<double>, pow_a<double>

// phi_ref (nullable, n x ncones): cone distances of every row at the current iterate; a trial row with
// phi < frac*phi_ref is treated as infeasible (fraction-to-the-boundary rule of the line search).
__global__ __launch_bounds__(kBlock) void barrier_f0_kernel(int n, BarrierParams P, const double* __restrict__ Dz,
                                                             const double* __restrict__ w,
                                                             const double* __restrict__ c,
                                                             const double* __restrict__ phi_ref, double frac,
                                                             double* __restrict__ phi_out, double* scratch,
                                                             double* out_dev, double* out_host, HostSignal sig) {
  __shared__ double lds[kBlock / 64];
  double accF = 0.0, accL = 0.0;
  for (long long q = (long long)blockIdx.x * kBlock + threadIdx.x; q < n; q += (long long)gridDim.x * kBlock) {
    const double* dz = Dz + q * P.K;
    const double* cq = c + q * P.K;
    const double wq = w[q];
    double F = 0.0;
    for (int ci = 0; ci < P.ncones; ++ci) {
      if (!P.active(ci, q)) {      // inactive piece: no constraint at this node
        if (phi_out) phi_out[q * P.ncones + ci] = INFINITY;
        continue;
      }
      const ConeAM<double> am = cone_am<double>(P, ci, q);
      Cone k = load_cone<double>(P.cone[ci], dz, am.a);
      if (phi_ref && !(k.phi >= frac * phi_ref[q * P.ncones + ci])) k.ok = false;
      if (phi_out) phi_out[q * P.ncones + ci] = k.phi;
      F += k.ok ? (-log(k.phi) - am.mu * log(k.s)) : INFINITY;
    }
    accF += wq * F;
    double lin = 0.0;
    for (int j = 0; j < P.K; ++j) lin += cq[j] * dz[j];
    accL += wq * lin;
  }
  const double r[2] = {block_sum(accF, lds), block_sum(accL, lds)};
  grid_finish<2>(r, scratch, out_dev, out_host, lds, sig);
}


// One objective evaluation of the line search in ONE launch (reference: f0 = apply_D, then map_rows of the barrier,
// then two dots -- SURVEY 8a a3/a4): x = s + alpha * nstep is formed on the fly (and written to s_out for the
// caller that accepts the trial), Dz = Dz0 + B x is computed for 64 nodes (64 K consecutive rows of B) per pass with
// the lane layout and summation order of spmv_kernel<G> -- bitwise the same Dz -- kept in LDS and written once, and
// the barrier terms of those nodes are accumulated straight from LDS.  Saves the waxpby and barrier_f0 launches and
// the re-read of Dz; bytes = spmv(B) + n (K + 2 + ncones [+ ncones]) 8.
constexpr int kTrialNodes = 64;
// NA = 1..3 trial points x_a = s + alpha_a * nstep share ONE pass over B (each nonzero is read once and feeds NA
// accumulators; per point the operations and their order are exactly those of the NA = 1 kernel, so every Dz and every sum
// is bitwise what separate launches give).  The barrier phase runs the points side by side: wave a takes point a.
template <int G, int NA>
__global__ __launch_bounds__(kBlock) void trial_f0_kernel(int n, int N, BarrierParams P, const int* __restrict__ rowptr,
                                                           const int* __restrict__ colidx, const double* __restrict__ vals,
                                                           const double* __restrict__ s, const double* __restrict__ nstep,
                                                           TrialSet T, const double* __restrict__ Dz0,
                                                           const double* __restrict__ w, const double* __restrict__ c,
                                                           const double* __restrict__ phi_ref, double frac, double* scratch,
                                                           HostSignal sig) {
  __shared__ double lds[kBlock / 64];
  __shared__ double dzs[NA][kTrialNodes * kMaxK];
  constexpr int GR = kBlock / G;      // rows per pass
  const int K = P.K, lane = threadIdx.x % G, grp = threadIdx.x / G;
#pragma unroll
  for (int a = 0; a < NA; ++a)
    if (T.s_out[a])
      for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < N; i += (long long)gridDim.x * kBlock)
        T.s_out[a][i] = s[i] + T.alpha[a] * nstep[i];
  auto xval = [&](int j, int a) { return nstep ? s[j] + T.alpha[a] * nstep[j] : s[j]; };
  double accF = 0.0, accL = 0.0;      // of the point this thread's wave evaluates
  const int pa = threadIdx.x >> 6, pnode = threadIdx.x & 63;
  const int nchunks = (n + kTrialNodes - 1) / kTrialNodes;
  for (int ch = xcd_block(blockIdx.x, gridDim.x); ch < nchunks; ch += gridDim.x) {
    const int q0 = ch * kTrialNodes, nq = min(kTrialNodes, n - q0), nrows = nq * K;
    const long long r0 = (long long)q0 * K;
    for (int rr0 = grp; rr0 < nrows; rr0 += kSpmvU * GR) {
      int b[kSpmvU], e[kSpmvU], ci[kSpmvU];
      double acc[NA][kSpmvU], base[kSpmvU], va[kSpmvU];
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
      for (int a = 0; a < NA; ++a)
#pragma unroll
        for (int u = 0; u < kSpmvU; ++u) acc[a][u] = (ci[u] >= 0) ? va[u] * xval(ci[u], a) : 0.0;
#pragma unroll
      for (int u = 0; u < kSpmvU; ++u)
        for (int k = b[u] + lane + G; k < e[u]; k += G) {
          const double vk = vals[k];
          const int ck = colidx[k];
#pragma unroll
          for (int a = 0; a < NA; ++a) acc[a][u] += vk * xval(ck, a);
        }
#pragma unroll
      for (int a = 0; a < NA; ++a)
#pragma unroll
        for (int u = 0; u < kSpmvU; ++u) {
          double t = acc[a][u];
#pragma unroll
          for (int o = G / 2; o > 0; o >>= 1) t += __shfl_down(t, o, G);
          const int rr = rr0 + u * GR;
          if (lane == 0 && rr < nrows) {
            const double v = base[u] + t;
            T.dz[a][r0 + rr] = v;
            dzs[a][rr] = v;
          }
        }
    }
    __syncthreads();
    if (pa < NA && pnode < nq) {      // wave-uniform in pa
      const long long q = q0 + pnode;
      const double* dz = dzs[pa] + pnode * K;
      const double* cq = c + q * K;
      const double wq = w[q];
      double* phi_out = T.phi_out[pa];
      double F = 0.0;
      for (int ci2 = 0; ci2 < P.ncones; ++ci2) {
        if (!P.active(ci2, q)) {
          if (phi_out) phi_out[q * P.ncones + ci2] = INFINITY;
          continue;
        }
        const ConeAM<double> am = cone_am<double>(P, ci2, q);
        Cone k = load_cone<double>(P.cone[ci2], dz, am.a);
        if (phi_ref && !(k.phi >= frac * phi_ref[q * P.ncones + ci2])) k.ok = false;
        if (phi_out) phi_out[q * P.ncones + ci2] = k.phi;
        F += k.ok ? (-log(k.phi) - am.mu * log(k.s)) : INFINITY;
      }
      accF += wq * F;
      double lin = 0.0;
      for (int j = 0; j < K; ++j) lin += cq[j] * dz[j];
      accL += wq * lin;
    }
    __syncthreads();
  }
  // per point: only its wave holds non-zero terms, so the block sums are those of the one-point kernel
  double r[2 * NA], mine[2 * NA];
#pragma unroll
  for (int a = 0; a < NA; ++a) {
    mine[2 * a] = pa == a ? accF : 0.0;
    mine[2 * a + 1] = pa == a ? accL : 0.0;
  }
  block_sum_n<2 * NA>(mine, r);      // one pair of barriers for all the points' sums
  // the partials go to the slot of the chunk sequence this block worked on, so the sums are those of the natural order;
  // results land at T.out_dev[0] .. (2 NA consecutive doubles: the points' (F, c.Dz) pairs) and its pinned host twin
  grid_finish<2 * NA>(r, scratch, T.out_dev, T.out_host, lds, sig, xcd_block(blockIdx.x, gridDim.x));
}

inline int trial_grid(int n) {
  const int b = (n + kTrialNodes - 1) / kTrialNodes;
  return b < 1 ? 1 : (b > kMaxBlocks ? kMaxBlocks : b);
}

template <int G>
void trial_launch(hipStream_t st, const DevCsr& B, int n, const BarrierParams& P, const double* s, const double* nstep,
                  const TrialSet& T, const double* Dz0, const double* w, const double* c, const double* phi_ref, double frac,
                  double* scratch, HostSignal sig) {
#define MGB_TRIAL_NA(NA)                                                                                                    \
  hipLaunchKernelGGL((trial_f0_kernel<G, NA>), dim3(trial_grid(n)), dim3(kBlock), 0, st, n, B.cols, P, B.rowptr, B.colidx,   \
                     B.vals, s, nstep, T, Dz0, w, c, phi_ref, frac, scratch, sig)
  switch (T.na) {
    case 1: MGB_TRIAL_NA(1); break;
    case 2: MGB_TRIAL_NA(2); break;
    default: MGB_TRIAL_NA(3); break;
  }
#undef MGB_TRIAL_NA
}

// map_rows of the barrier itself (src:161-170 for the closures MultiGridBarrier builds from a convex set): F per row, and the
// per-row Hessian in the reference's flattened K x K shape (column (j-1) K + k, test/test_map_rows_compare.jl:73) expanded
// from the packed slots of barrier_f2_kernel.
__global__ __launch_bounds__(kBlock) void barrier_rows_F_kernel(int n, BarrierParams P, const double* __restrict__ Dz,
                                                                 double* __restrict__ out) {
  for (long long q = (long long)blockIdx.x * kBlock + threadIdx.x; q < n; q += (long long)gridDim.x * kBlock) {
    const double* dz = Dz + q * P.K;
    double F = 0.0;
    for (int ci = 0; ci < P.ncones; ++ci) {
      if (!P.active(ci, q)) continue;
      const ConeAM<double> am = cone_am<double>(P, ci, q);
      Cone k = load_cone<double>(P.cone[ci], dz, am.a);
      F += k.ok ? (-log(k.phi) - am.mu * log(k.s)) : INFINITY;
    }
    out[q] = F;
  }
}

__global__ __launch_bounds__(kBlock) void expand_hessian_rows_kernel(int n, BarrierParams P, const double* __restrict__ Y,
                                                                      double* __restrict__ out) {
  const int K = P.K, nY = P.nY();
  for (long long q = (long long)blockIdx.x * kBlock + threadIdx.x; q < n; q += (long long)gridDim.x * kBlock) {
    double* o = out + q * K * K;
    for (int e = 0; e < K * K; ++e) o[e] = 0.0;
    const double* yq = Y + q * nY;
    int slot = 0;
    for (int ci = 0; ci < P.ncones; ++ci) {
      const ConeSpec& S = P.cone[ci];
      const int nact = S.nact();
      for (int a = 0; a < nact; ++a)
        for (int b = a; b < nact; ++b, ++slot) {
          const int ra = S.col(a), rb = S.col(b);
          o[ra * K + rb] += yq[slot];
          if (ra != rb) o[rb * K + ra] += yq[slot];
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

// flag_dev / flag_host (nullable): the pivot flag of the factorisation that produced y -- all its launches precede this
// one on the stream -- is handed to the host next to the dot product and re-armed (zeroed) for the next factorisation, so the
// Newton loop needs neither a memset nor a copy launch for it.
__global__ __launch_bounds__(kBlock) void dot_kernel(int n, const double* __restrict__ x, const double* __restrict__ y,
                                                      double* scratch, double* out_dev, double* out_host, int* flag_dev,
                                                      int* flag_host, HostSignal sig) {
  __shared__ double lds[kBlock / 64];
  if (flag_dev && blockIdx.x == 0 && threadIdx.x == 0) {
    const int f = __hip_atomic_load(flag_dev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(flag_host, f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(flag_dev, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  double acc = 0.0;
  for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < n; i += (long long)gridDim.x * kBlock)
    acc += x[i] * y[i];
  const double r[1] = {block_sum(acc, lds)};
  grid_finish<1>(r, scratch, out_dev, out_host, lds, sig);
}

__global__ __launch_bounds__(kBlock) void dot_owned_kernel(int n, const double* __restrict__ x, const double* __restrict__ y,
                                                            const int* __restrict__ kind, int count_top, double* scratch,
                                                            double* out_dev) {
  __shared__ double lds[kBlock / 64];
  double acc = 0.0;
  for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < n; i += (long long)gridDim.x * kBlock) {
    const int k = kind[i];
    if (k == 1 || (k == 2 && count_top)) acc += x[i] * y[i];      // select, never multiply by a mask: the other entries may hold anything
  }
  const double r[1] = {block_sum(acc, lds)};
  grid_finish<1>(r, scratch, out_dev, nullptr, lds);
}

__global__ __launch_bounds__(kBlock) void gather_top_kernel(int ntop, const int* __restrict__ top, const double* __restrict__ g,
                                                             double* __restrict__ buf) {
  for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < ntop; i += (long long)gridDim.x * kBlock) buf[i] = g[top[i]];
}

// one workgroup: scatter the summed top entries back and finish |g|^2 in a fixed order
__global__ __launch_bounds__(kBlock) void scatter_top_norm_kernel(int ntop, const int* __restrict__ top, const double* __restrict__ buf,
                                                                   double* __restrict__ g, double* out) {
  __shared__ double lds[kBlock / 64];
  double acc = 0.0;
  for (int i = threadIdx.x; i < ntop; i += kBlock) {
    const double v = buf[i];
    g[top[i]] = v;
    acc += v * v;
  }
  const double tot = block_sum(acc, lds);
  if (threadIdx.x == 0) out[0] = buf[ntop] + tot;
}

__global__ void flag_to_double_kernel(const int* flag_dev, double* out) {
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = *flag_dev != 0 ? 1.0 : 0.0;
}

__global__ __launch_bounds__(kBlock) void sum_kernel(int n, const double* __restrict__ x, double* scratch, double* out_dev,
                                                      double* out_host) {
  __shared__ double lds[kBlock / 64];
  double acc = 0.0;
  for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < n; i += (long long)gridDim.x * kBlock) acc += x[i];
  const double r[1] = {block_sum(acc, lds)};
  grid_finish<1>(r, scratch, out_dev, out_host, lds);
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

void launch_spmv_el(hipStream_t st, const DevCsr& A, const DevElCsr& E, const double* x, const double* y0, double* y) {
  if (A.rows == 0) return;
  switch (A.group) {
    case 1: spmv_el_launch<1>(st, A, E, x, y0, y); break;
    case 2: spmv_el_launch<2>(st, A, E, x, y0, y); break;
    case 4: spmv_el_launch<4>(st, A, E, x, y0, y); break;
    case 8: spmv_el_launch<8>(st, A, E, x, y0, y); break;
    case 16: spmv_el_launch<16>(st, A, E, x, y0, y); break;
    case 32: spmv_el_launch<32>(st, A, E, x, y0, y); break;
    default: spmv_el_launch<64>(st, A, E, x, y0, y); break;
  }
}

void launch_waxpby(hipStream_t st, int n, const double* x, double alpha, const double* y, double* out) {
  if (n == 0) return;
  hipLaunchKernelGGL(waxpby_kernel, dim3(grid_for(n)), dim3(kBlock), 0, st, n, x, alpha, y, out);
}

int f0_blocks(int n) { return std::max(grid_for(n), trial_grid(n)); }

void launch_trial_set(hipStream_t st, const DevCsr& B, int n, BarrierParams P, const double* s, const double* nstep,
                      const TrialSet& T, const double* Dz0, const double* w, const double* c, const double* phi_ref, double frac,
                      double* scratch, HostSignal sig) {
#define MGB_TRIAL(G) trial_launch<G>(st, B, n, P, s, nstep, T, Dz0, w, c, phi_ref, frac, scratch, sig)
  switch (B.group) {
    case 1: MGB_TRIAL(1); break;
    case 2: MGB_TRIAL(2); break;
    case 4: MGB_TRIAL(4); break;
    case 8: MGB_TRIAL(8); break;
    case 16: MGB_TRIAL(16); break;
    case 32: MGB_TRIAL(32); break;
    default: MGB_TRIAL(64); break;
  }
#undef MGB_TRIAL
}

void launch_trial_f0(hipStream_t st, const DevCsr& B, int n, BarrierParams P, const double* s, double alpha,
                     const double* nstep, double* s_out, const double* Dz0, double* Dz, const double* w, const double* c,
                     const double* phi_ref, double frac, double* phi_out, double* scratch, double* out2, double* out2_host,
                     HostSignal sig) {
  TrialSet T;
  T.na = 1;
  T.alpha[0] = alpha;
  T.s_out[0] = s_out;
  T.dz[0] = Dz;
  T.phi_out[0] = phi_out;
  T.out_dev = out2;
  T.out_host = out2_host;
  launch_trial_set(st, B, n, P, s, nstep, T, Dz0, w, c, phi_ref, frac, scratch, sig);
}

void launch_barrier_f0(hipStream_t st, int n, BarrierParams P, const double* Dz, const double* w, const double* c,
                       const double* phi_ref, double frac, double* phi_out, double* scratch, double* out2, double* out2_host,
                       HostSignal sig) {
  hipLaunchKernelGGL(barrier_f0_kernel, dim3(grid_for(n)), dim3(kBlock), 0, st, n, P, Dz, w, c, phi_ref, frac, phi_out, scratch,
                     out2, out2_host, sig);
}

void launch_barrier_f1(hipStream_t st, int n, BarrierParams P, const double* Dz, const double* w, const double* c,
                       double t, double* v) {
  hipLaunchKernelGGL(barrier_f1_kernel_t<double>, dim3(grid_for(n)), dim3(kBlock), 0, st, n, P, Dz, w, c, t, v);
}

void launch_barrier_f2(hipStream_t st, int n, BarrierParams P, const double* Dz, const double* w, double* Y) {
  hipLaunchKernelGGL(barrier_f2_kernel_t<double>, dim3(grid_for(n)), dim3(kBlock), 0, st, n, P, Dz, w, Y);
}

void launch_barrier_rows_F(hipStream_t st, int n, BarrierParams P, const double* Dz, double* out) {
  if (n == 0) return;
  hipLaunchKernelGGL(barrier_rows_F_kernel, dim3(grid_for(n)), dim3(kBlock), 0, st, n, P, Dz, out);
}

void launch_expand_hessian_rows(hipStream_t st, int n, BarrierParams P, const double* Y, double* out) {
  if (n == 0) return;
  hipLaunchKernelGGL(expand_hessian_rows_kernel, dim3(grid_for(n)), dim3(kBlock), 0, st, n, P, Y, out);
}

void launch_dot(hipStream_t st, int n, const double* x, const double* y, double* scratch, double* out, double* out_host,
                int* flag_dev, int* flag_host, HostSignal sig) {
  hipLaunchKernelGGL(dot_kernel, dim3(grid_for(n)), dim3(kBlock), 0, st, n, x, y, scratch, out, out_host, flag_dev, flag_host, sig);
}

void launch_dot_owned(hipStream_t st, int n, const double* x, const double* y, const int* kind, int count_top, double* scratch,
                      double* out) {
  hipLaunchKernelGGL(dot_owned_kernel, dim3(grid_for(n)), dim3(kBlock), 0, st, n, x, y, kind, count_top, scratch, out);
}

void launch_gather_top(hipStream_t st, int ntop, const int* top, const double* g, double* buf) {
  if (ntop > 0) hipLaunchKernelGGL(gather_top_kernel, dim3(grid_for(ntop)), dim3(kBlock), 0, st, ntop, top, g, buf);
}

void launch_scatter_top_norm(hipStream_t st, int ntop, const int* top, const double* buf, double* g, double* out) {
  hipLaunchKernelGGL(scatter_top_norm_kernel, dim3(1), dim3(kBlock), 0, st, ntop, top, buf, g, out);
}

void launch_flag_to_double(hipStream_t st, const int* flag_dev, double* out) {
  hipLaunchKernelGGL(flag_to_double_kernel, dim3(1), dim3(64), 0, st, flag_dev, out);
}

void launch_sum(hipStream_t st, int n, const double* x, double* scratch, double* out, double* out_host) {
  hipLaunchKernelGGL(sum_kernel, dim3(grid_for(n)), dim3(kBlock), 0, st, n, x, scratch, out, out_host);
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
