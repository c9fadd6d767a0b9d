// gfx950 (CDNA4, wave64) kernels of the multigrid pieces (mg.hpp): matrix-free Hessian product,
// Chebyshev-Jacobi smoothing steps, dense coarsest-level inverse, device-resident CG scalars.
//
// Roofline: everything here is HBM / L2-bandwidth bound, fp64, AI < 1 flop/B; no MFMA (sparse / irregular, not a contraction).
//   elop_apply   bytes = nnz(B) * 8 (values, read ONCE for both halves B v and B' u) + nel * cmax * 4 (column ids) + nel * 4 (class)
//                        + n * nY * 8 (Y) + nel * cmax * 8 (element results) + gathered vin  (class tables stay cache resident)
//   dof_gather   bytes = nel * cmax * (8 + 4) + (N + 1) * 4 + epilogue vectors (2 .. 6 N doubles)
//   csr_apply<G> bytes = nnz * 12 + (N + 1) * 4 + epilogue vectors
// against plain CSR products for the same H v (B: nnz * 12, B': nnz * 12, Dz and u round trips through memory).
#include "mg.hpp"

#include <algorithm>
#include <cmath>
#include <vector>

#include "devutil.hpp"

namespace mgb {

namespace {

__device__ inline bool mg_done(const double* done) { return done && *done != 0.0; }

// Per D-row a: the (b, slot) pairs with a nonzero d2F / dDz_a dDz_b, over all barrier terms (barrier_f2_kernel's packed upper
// triangles, mirrored): u_a = sum over a's pairs of Y[slot] d_b.  Uniform over the nodes.
struct YList {
  unsigned char ptr[9];
  unsigned short pair[8 * 8 * kMaxCones];      // b | slot << 8
};

// Phase 1 of the matrix-free product.  TPE threads work on one element, kBlock / TPE elements per pass; everything an element
// needs is staged in LDS by ONE round of independent, coalesced loads (class id -> its structure table, column ids -> vin,
// nonzero values, Y), then LDS-only phases:
//   C   ds <- B_e xs           lane = row
//   D   us <- Y_q ds           lane = row (node q, D-row a) through the pair list of a
//   E1  pr <- B_e(k) us        lane = nonzero, in column-wise order (balanced whatever the column lengths are)
//   E2  elbuf <- column sums   lane = element column; rows ascending within a column: fixed order, reproducible
// diag != 0: E forms diag(B_e' Y_e B_e) instead.
// Class table entry per nonzero (64 bit): bits 0-15 local nonzero index and 16-31 local row of the p-th entry in column-wise
// order, bits 32-39 local column of the k-th entry in row-wise order.
constexpr int kPre = 4;      // register prefetch depth per lane (values / table-free data of the NEXT pass); deeper elements take the plain path

template <int TPE>
__global__ __launch_bounds__(kBlock) void elop_apply_kernel(DevElOp E, int K, int nY, YList yl, const double* __restrict__ Y,
                                                             const double* __restrict__ v, const double* __restrict__ vmul,
                                                             const double* __restrict__ vscale, double* __restrict__ elbuf,
                                                             int diag, const double* done) {
  if (mg_done(done)) return;
  extern __shared__ double lds_el[];
  __shared__ YList yls;
  constexpr int EPB = kBlock / TPE;
  const int rpe = E.rows_per_el, cmax = E.cmax, nzm = E.nnz_max, blk = E.block;
  const int el = threadIdx.x / TPE, ln = threadIdx.x % TPE;
  if (threadIdx.x < 9) yls.ptr[threadIdx.x] = yl.ptr[threadIdx.x];
  for (int i = threadIdx.x; i < 8 * 8 * kMaxCones; i += kBlock) yls.pair[i] = yl.pair[i];
  // this element's LDS slot
  double* base = lds_el + (size_t)el * E.slot_doubles;
  double* xs = base;
  double* vs = xs + cmax;
  double* ds = vs + nzm;
  double* us = ds + rpe;
  double* ys = us + rpe;
  double* pr = ys + (size_t)blk * nY;
  unsigned long long* ent = reinterpret_cast<unsigned long long*>(pr + nzm);
  unsigned short* rp = reinterpret_cast<unsigned short*>(ent + nzm);
  unsigned short* tp = rp + (rpe + 1);
  const unsigned char* entb = reinterpret_cast<const unsigned char*>(ent);
  const unsigned* entw = reinterpret_cast<const unsigned*>(ent);
  const double vsc = vscale ? *vscale : 1.0;
  const int npass = (E.nel + EPB - 1) / EPB;
  const int nYel = blk * nY;
  // Software pipeline: the HBM-latency loads of pass n + 1 (nonzero values, Y, column ids, class, extent) are issued into
  // registers before the LDS-only phases of pass n and land in LDS at the top of pass n + 1; only the gather of vin (its column
  // ids are already in registers) and a class table that changed are fetched inside the pass.
  const bool pipelined = (nzm + TPE - 1) / TPE <= kPre && (nYel + TPE - 1) / TPE <= kPre && (cmax + TPE - 1) / TPE <= kPre;
  double pv[kPre], py[kPre];
  int pcol[kPre], pc = 0, pk0 = 0, pnz = 1, pe = 0;      // stage B (next pass): values, Y, column ids
  int qc = 0, qk0 = 0, qnz = 1, qe = 0;                   // stage A (the pass after): class and extent, which stage B's addresses need
  // both stages load unconditionally at clamped (always valid) addresses -- a condition per load makes the compiler branch
  // around it and wait for each one in turn; passes beyond the end recompute the last element and never store
  auto stage_a = [&](int ps) {
    qe = min(ps * EPB + el, E.nel - 1);
    qc = E.cls[qe];
    qk0 = E.rowptr[(size_t)qe * rpe];
    qnz = E.rowptr[(size_t)(qe + 1) * rpe] - qk0;
  };
  auto stage_b = [&]() {
    pe = qe;
    pc = qc;
    pk0 = qk0;
    pnz = qnz;
#pragma unroll
    for (int u = 0; u < kPre; ++u) pcol[u] = E.ecols[(size_t)pe * cmax + min(ln + u * TPE, cmax - 1)];
#pragma unroll
    for (int u = 0; u < kPre; ++u) pv[u] = E.vals[pk0 + min(ln + u * TPE, pnz - 1)];
#pragma unroll
    for (int u = 0; u < kPre; ++u) py[u] = Y[(size_t)pe * nYel + min(ln + u * TPE, nYel - 1)];
  };
  int have_cls = -1;      // class whose table this element slot holds: neighbouring elements mostly share it, so it is loaded once
  int ps = xcd_block(blockIdx.x, gridDim.x);
  if (pipelined && ps < npass) {
    stage_a(ps);
    stage_b();
    stage_a(ps + gridDim.x);
  }
  for (; ps < npass; ps += gridDim.x) {
    const int e = ps * EPB + el;
    bool live = e < E.nel;
    int nz = 0;
    __syncthreads();      // the previous pass is done with the staging buffers
    if (pipelined) {
      nz = pnz;
      const int c = pc;
      double xg[kPre];
#pragma unroll
      for (int u = 0; u < kPre; ++u) xg[u] = diag ? 0.0 : v[pcol[u]];
      if (vmul && !diag) {
#pragma unroll
        for (int u = 0; u < kPre; ++u) xg[u] *= vmul[pcol[u]];
      }
#pragma unroll
      for (int u = 0; u < kPre; ++u) {
        const int k = ln + u * TPE;
        if (k < nz) vs[k] = pv[u];
        if (k < nYel) ys[k] = py[u];
      }
      // the next pass's loads go out before this pass's arithmetic (their class / extent arrived during the previous pass)
      stage_b();
      stage_a(ps + 2 * gridDim.x);
#pragma unroll
      for (int u = 0; u < kPre; ++u) {
        const int j = ln + u * TPE;
        if (j < cmax) xs[j] = vsc * xg[u];
      }
      if (c != have_cls) {      // uniform over the element's lanes; rare: neighbouring elements share their class
        for (int i = ln; i <= rpe; i += TPE) rp[i] = E.c_rowptr[(size_t)c * (rpe + 1) + i];
        for (int i = ln; i <= cmax; i += TPE) tp[i] = E.c_tptr[(size_t)c * (cmax + 1) + i];
        for (int i = ln; i < nz; i += TPE) ent[i] = E.c_ent[(size_t)c * nzm + i];
        have_cls = c;
      }
    } else if (live) {
      const int c = E.cls[e];
      const int k0 = E.rowptr[(size_t)e * rpe];
      nz = E.rowptr[(size_t)(e + 1) * rpe] - k0;
      if (!diag)
        for (int j = ln; j < cmax; j += TPE) {
          const int col = E.ecols[(size_t)e * cmax + j];
          double x = v[col];
          if (vmul) x *= vmul[col];
          xs[j] = vsc * x;
        }
      for (int k = ln; k < nz; k += TPE) vs[k] = E.vals[k0 + k];
      for (int i = ln; i < nYel; i += TPE) ys[i] = Y[(size_t)e * nYel + i];
      if (c != have_cls) {      // uniform over the element's lanes
        for (int i = ln; i <= rpe; i += TPE) rp[i] = E.c_rowptr[(size_t)c * (rpe + 1) + i];
        for (int i = ln; i <= cmax; i += TPE) tp[i] = E.c_tptr[(size_t)c * (cmax + 1) + i];
        for (int i = ln; i < nz; i += TPE) ent[i] = E.c_ent[(size_t)c * nzm + i];
        have_cls = c;
      }
    }
    __syncthreads();
    if (!diag) {
      if (live)
        for (int rr = ln; rr < rpe; rr += TPE) {
          double acc = 0.0;
          for (int k = rp[rr]; k < rp[rr + 1]; ++k) acc += vs[k] * xs[entb[8 * k + 4]];
          ds[rr] = acc;
        }
      __syncthreads();
      if (live)
        for (int rr = ln; rr < rpe; rr += TPE) {
          const int node = rr / K, a = rr - node * K;
          const double* yq = ys + node * nY;
          const double* dq = ds + node * K;
          double acc = 0.0;
          for (int i = yls.ptr[a]; i < yls.ptr[a + 1]; ++i) {
            const unsigned pq = yls.pair[i];
            acc += yq[pq >> 8] * dq[pq & 0xffu];
          }
          us[rr] = acc;
        }
      __syncthreads();
      if (live)
        for (int p = ln; p < nz; p += TPE) {
          const unsigned w = entw[2 * p];
          pr[p] = vs[w & 0xffffu] * us[w >> 16];
        }
      __syncthreads();
      if (live)
        for (int j = ln; j < cmax; j += TPE) {
          double acc = 0.0;
          for (int p = tp[j]; p < tp[j + 1]; ++p) acc += pr[p];
          elbuf[(size_t)e * cmax + j] = acc;
        }
    } else if (live) {
      // diagonal entry of column j: sum over pairs of the column's entries that sit on the same node of  b1 Y_q[a1, a2] b2
      for (int j = ln; j < cmax; j += TPE) {
        double acc = 0.0;
        for (int p1 = tp[j]; p1 < tp[j + 1]; ++p1) {
          const unsigned w1 = entw[2 * p1];
          const int r1 = w1 >> 16, node = r1 / K, a1 = r1 - node * K;
          const double* yq = ys + node * nY;
          for (int p2 = tp[j]; p2 < tp[j + 1]; ++p2) {
            const unsigned w2 = entw[2 * p2];
            const int r2 = w2 >> 16;
            if (r2 / K != node) continue;
            const unsigned a2 = r2 - node * K;
            double y = 0.0;
            for (int i = yls.ptr[a1]; i < yls.ptr[a1 + 1]; ++i) {
              const unsigned pq = yls.pair[i];
              y += (pq & 0xffu) == a2 ? yq[pq >> 8] : 0.0;
            }
            acc += vs[w1 & 0xffffu] * y * vs[w2 & 0xffffu];
          }
        }
        elbuf[(size_t)e * cmax + j] = acc;
      }
    }
  }
}

// Element matrices for the assembly (mg.hpp: DevElAsm).  The element's nonzero values and its Y are staged in LDS; one lane per
// structurally nonzero pair (i >= j) then sums the pair's products  v[k1] * Y[node, slot] * v[k2]  listed once per structure class
// (the class-level restriction of the reference's recipe, test/test_map_rows_compare.jl:111-122): no searching, fixed order.
template <int TPE>
__global__ __launch_bounds__(kBlock) void elop_assemble_kernel(DevElOp E, DevElAsm A, int nY, const double* __restrict__ Y,
                                                                double* __restrict__ elmat) {
  extern __shared__ double lds_el[];
  constexpr int EPB = kBlock / TPE;
  const int rpe = E.rows_per_el, nzm = E.nnz_max, nYel = E.block * nY;
  const int el = threadIdx.x / TPE, ln = threadIdx.x % TPE;
  double* vs = lds_el + (size_t)el * A.slot_doubles;
  double* ys = vs + nzm;
  const int npass = (E.nel + EPB - 1) / EPB;
  for (int ps = xcd_block(blockIdx.x, gridDim.x); ps < npass; ps += gridDim.x) {
    const int e = ps * EPB + el;
    const bool live = e < E.nel;
    __syncthreads();
    if (live) {
      const int k0 = E.rowptr[(size_t)e * rpe], nz = E.rowptr[(size_t)(e + 1) * rpe] - k0;
      for (int k = ln; k < nz; k += TPE) vs[k] = E.vals[k0 + k];
      for (int i = ln; i < nYel; i += TPE) ys[i] = Y[(size_t)e * nYel + i];
    }
    __syncthreads();
    if (live) {
      const int c = E.cls[e], np = A.c_npairs[c];
      const int* tptr = A.c_tptr + (size_t)c * (A.npm + 1);
      const unsigned long long* terms = A.c_terms + (size_t)c * A.ntm;
      for (int s = ln; s < np; s += TPE) {
        double acc = 0.0;
        for (int t = tptr[s]; t < tptr[s + 1]; ++t) {
          const unsigned long long w = terms[t];
          acc += vs[w & 0xffffu] * ys[(w >> 32) & 0xffffu] * vs[(w >> 16) & 0xffffu];
        }
        elmat[(size_t)e * A.npm + s] = acc;
      }
    }
  }
}

// out[r] = sum_{k in ptr[r] .. ptr[r + 1]} in[idx[k]], fixed order
__global__ __launch_bounds__(kBlock) void gather_sum_kernel(int n, const int* __restrict__ ptr, const int* __restrict__ idx,
                                                             const double* __restrict__ in, double* __restrict__ out) {
  for (long long r = (long long)blockIdx.x * kBlock + threadIdx.x; r < n; r += (long long)gridDim.x * kBlock) {
    double t = 0.0;
    for (int k = ptr[r]; k < ptr[r + 1]; ++k) t += in[idx[k]];
    out[r] = t;
  }
}

__device__ inline double mg_epilogue(const MgEpi& E, int i, double t) {
  switch (E.mode) {
    case MG_PLAIN:
      E.out[i] = t;
      return 0.0;
    case MG_FIRST: {
      const double di = E.dinv[i], bi = E.b[i];
      const double vi = E.coef[0] * di * bi;
      const double ri = bi - t;
      E.x[i] = vi;
      E.r[i] = ri;
      if (E.has_next) E.d_new[i] = E.coef[2 * E.k - 1] * vi + E.coef[2 * E.k] * di * ri;
      return 0.0;
    }
    case MG_STEP: {
      const double dold = E.v[i];
      double xi = E.x[i] + dold;
      const double ri = E.r[i] - t;
      E.r[i] = ri;
      if (E.has_next) {
        const double dn = E.coef[2 * E.k - 1] * dold + E.coef[2 * E.k] * E.dinv[i] * ri;
        E.d_new[i] = dn;
        if (E.add_new) xi += dn;
      }
      E.x[i] = xi;
      return 0.0;
    }
    case MG_RESID: {
      const double ri = E.b[i] - t;
      E.r[i] = ri;
      E.d_new[i] = E.coef[0] * E.dinv[i] * ri;
      return 0.0;
    }
    case MG_PAP:
      E.out[i] = t;
      return E.v[i] * t;
    case MG_POWER: {
      const double y = E.dinv[i] * t;
      E.out[i] = y;
      return y * t;      // y' D y
    }
    default:      // MG_DIAGINV
      E.out[i] = 1.0 / t;
      return 0.0;
  }
}

// thread 0 of the last block: turn the launch's sum into the scalars the next launches read
__device__ inline void mg_finish_scalars(const MgEpi& E, double tot) {
  double* s = E.scal;
  if (E.mode == MG_PAP) {
    s[SC_PAP] = tot;
    if (!(tot > 0.0) || !isfinite(tot)) s[SC_DONE] = 2.0;      // breakdown: H is not positive definite along p
    else s[SC_ALPHA] = s[SC_RZ] / tot;
  } else {      // MG_POWER: |Dinv A v|_D with |v|_D = 1
    const double lam = sqrt(tot);
    s[SC_LMAX] = lam;
    s[SC_VSCALE] = lam > 0.0 ? 1.0 / lam : 0.0;
  }
}

// Phase 2: out_i = sum of the element results of dof i (fixed order), then the epilogue of the launch
__global__ __launch_bounds__(kBlock) void dof_gather_kernel(DevElOp E, const double* __restrict__ elbuf, MgEpi epi) {
  if (mg_done(epi.done)) return;
  __shared__ double lds[kBlock / 64];
  double acc = 0.0;
  for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < E.N; i += (long long)gridDim.x * kBlock) {
    double t = 0.0;
    for (int p = E.dptr[i]; p < E.dptr[i + 1]; ++p) t += elbuf[E.didx[p]];
    acc += mg_epilogue(epi, (int)i, t);
  }
  if (epi.mode == MG_PAP || epi.mode == MG_POWER) {
    const double r[1] = {block_sum(acc, lds)};
    grid_finish_fn<1>(r, epi.scratch, lds, [&](const double (&tot)[1]) { mg_finish_scalars(epi, tot[0]); });
  }
}

// assembled level: G lanes per row as spmv_kernel<G> (fixed shuffle tree), epilogue by lane 0
template <int G>
__global__ __launch_bounds__(kBlock) void csr_apply_kernel(int rows, const int* __restrict__ rowptr, const int* __restrict__ colidx,
                                                            const double* __restrict__ vals, MgEpi epi) {
  if (mg_done(epi.done)) return;
  __shared__ double lds[kBlock / 64];
  const int lane = threadIdx.x % G;
  const double vsc = epi.vscale ? *epi.vscale : 1.0;
  double racc = 0.0;
  const long long stride = (long long)gridDim.x * (kBlock / G);
  for (long long row = (long long)xcd_block(blockIdx.x, gridDim.x) * (kBlock / G) + threadIdx.x / G; row < rows; row += stride) {
    double acc = 0.0;
    for (int k = rowptr[row] + lane; k < rowptr[row + 1]; k += G) {
      const int j = colidx[k];
      double x = epi.v[j];
      if (epi.vmul) x *= epi.vmul[j];
      acc += vals[k] * x;
    }
#pragma unroll
    for (int o = G / 2; o > 0; o >>= 1) acc += __shfl_down(acc, o, G);
    if (lane == 0) racc += mg_epilogue(epi, (int)row, vsc * acc);
  }
  if (epi.mode == MG_PAP || epi.mode == MG_POWER) {
    const double r[1] = {block_sum(racc, lds)};
    grid_finish_fn<1>(r, epi.scratch, lds, [&](const double (&tot)[1]) { mg_finish_scalars(epi, tot[0]); });
  }
}

__global__ __launch_bounds__(kBlock) void expand_sym_kernel(int nnz_full, const int* __restrict__ map, const double* __restrict__ lower,
                                                             double* __restrict__ full, int N, const int* __restrict__ diagpos,
                                                             double* __restrict__ dinv) {
  for (long long k = (long long)blockIdx.x * kBlock + threadIdx.x; k < nnz_full; k += (long long)gridDim.x * kBlock) full[k] = lower[map[k]];
  for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < N; i += (long long)gridDim.x * kBlock)
    dinv[i] = 1.0 / lower[map[diagpos[i]]];
}

__global__ void cheb_coef_kernel(const double* scal, double* coef, int degree, double lo_frac, double hi_frac) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const double lam = scal[SC_LMAX];
  const double hi = hi_frac * lam, lo = lo_frac * lam;
  const double theta = 0.5 * (hi + lo), delta = 0.5 * (hi - lo), sigma = theta / delta;
  double rho = 1.0 / sigma;
  coef[0] = 1.0 / theta;
  for (int k = 1; k < degree; ++k) {
    const double rn = 1.0 / (2.0 * sigma - rho);
    coef[2 * k - 1] = rn * rho;
    coef[2 * k] = 2.0 * rn / delta;
    rho = rn;
  }
  coef[kChebStride - 1] = lam;
}

// scal[SC_VSCALE] = 1 / |ev|_D
__global__ __launch_bounds__(kBlock) void power_start_kernel(int n, const double* __restrict__ ev, const double* __restrict__ dinv,
                                                              double* scal, double* scratch) {
  __shared__ double lds[kBlock / 64];
  double acc = 0.0;
  for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < n; i += (long long)gridDim.x * kBlock) acc += ev[i] * ev[i] / dinv[i];
  const double r[1] = {block_sum(acc, lds)};
  grid_finish_fn<1>(r, scratch, lds, [&](const double (&tot)[1]) { scal[SC_VSCALE] = tot[0] > 0.0 ? 1.0 / sqrt(tot[0]) : 0.0; });
}

// In-place Gauss-Jordan inversion of an SPD matrix in LDS (no pivoting: the pivots of an SPD matrix are positive), one workgroup.
// Row stride N + 1 (odd multiples of 8 bytes: column walks spread over the banks).
__global__ __launch_bounds__(kBlock) void dense_inverse_kernel(int N, const int* __restrict__ rowptr, const int* __restrict__ colidx,
                                                                const double* __restrict__ vals, double* __restrict__ Ainv, int* fail) {
  extern __shared__ double M[];
  __shared__ double pivinv;
  const int ld = N + 1;
  for (int idx = threadIdx.x; idx < N * ld; idx += kBlock) M[idx] = 0.0;
  __syncthreads();
  for (int i = threadIdx.x; i < N; i += kBlock)
    for (int k = rowptr[i]; k < rowptr[i + 1]; ++k) {
      const int j = colidx[k];
      M[i * ld + j] = vals[k];
      M[j * ld + i] = vals[k];
    }
  __syncthreads();
  for (int k = 0; k < N; ++k) {
    if (threadIdx.x == 0) {
      const double p = M[k * ld + k];
      if (!(p > 0.0) || !isfinite(p)) atomicOr(fail, 1);
      pivinv = 1.0 / p;
    }
    __syncthreads();
    const double pi = pivinv;
    // column k of the other rows is needed by every update of the step: keep it in M[i][N] (the padding column)
    for (int i = threadIdx.x; i < N; i += kBlock) M[i * ld + N] = M[i * ld + k];
    __syncthreads();
    for (int j = threadIdx.x; j < N; j += kBlock) M[k * ld + j] = (j == k) ? pi : M[k * ld + j] * pi;
    __syncthreads();
    for (int idx = threadIdx.x; idx < N * N; idx += kBlock) {
      const int i = idx / N, j = idx - i * N;
      if (i == k) continue;
      const double f = M[i * ld + N];
      M[i * ld + j] = (j == k) ? -f * pi : M[i * ld + j] - f * M[k * ld + j];
    }
    __syncthreads();
  }
  for (int idx = threadIdx.x; idx < N * N; idx += kBlock) {
    const int i = idx / N, j = idx - i * N;
    Ainv[idx] = 0.5 * (M[i * ld + j] + M[j * ld + i]);      // exactly symmetric: the V-cycle must be a symmetric operator
  }
}

// x = Ainv b: one wave per row, fixed shuffle tree
__global__ __launch_bounds__(kBlock) void dense_apply_kernel(int N, const double* __restrict__ Ainv, const double* __restrict__ b,
                                                              double* __restrict__ x, const double* done) {
  if (mg_done(done)) return;
  const int lane = threadIdx.x & 63;
  for (int row = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6); row < N; row += gridDim.x * (kBlock / 64)) {
    double acc = 0.0;
    for (int j = lane; j < N; j += 64) acc += Ainv[(size_t)row * N + j] * b[j];
    acc = wave_sum(acc);
    if (lane == 0) x[row] = acc;
  }
}

__global__ __launch_bounds__(kBlock) void pcg_init_kernel(int n, const double* __restrict__ g, double* __restrict__ x,
                                                           double* __restrict__ r, double* scal, double rtol, int maxit) {
  for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < n; i += (long long)gridDim.x * kBlock) {
    x[i] = 0.0;
    r[i] = g[i];
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    scal[SC_RZ] = 0.0;
    scal[SC_PAP] = 0.0;
    scal[SC_ALPHA] = 0.0;
    scal[SC_BETA] = 0.0;
    scal[SC_RZ0] = 0.0;
    scal[SC_TOL2] = rtol * rtol;
    scal[SC_ITER] = -1.0;      // the first dot launch (<r0, z0>) makes it 0
    scal[SC_DONE] = 0.0;
    scal[SC_MAXIT] = (double)maxit;
  }
}

__global__ __launch_bounds__(kBlock) void pcg_update_kernel(int n, double* __restrict__ x, double* __restrict__ r,
                                                             const double* __restrict__ p, const double* __restrict__ Ap,
                                                             const double* scal) {
  if (scal[SC_DONE] != 0.0) return;
  const double alpha = scal[SC_ALPHA];
  for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < n; i += (long long)gridDim.x * kBlock) {
    x[i] += alpha * p[i];
    r[i] -= alpha * Ap[i];
  }
}

__global__ __launch_bounds__(kBlock) void pcg_dot_kernel(int n, const double* __restrict__ r, const double* __restrict__ z, double* scal,
                                                          double* scratch, double* host4, HostSignal sig) {
  __shared__ double lds[kBlock / 64];
  auto signal = [&]() {      // thread 0 of one block
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long q = __hip_atomic_load(sig.seq_dev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1;
    __hip_atomic_store(sig.seq_dev, q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(sig.seq_host, q, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  };
  if (scal[SC_DONE] != 0.0) {      // finished earlier in this batch: nothing moves, the host still gets its signal
    if (sig.seq_host && blockIdx.x == 0 && threadIdx.x == 0) signal();
    return;
  }
  double acc = 0.0;
  for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < n; i += (long long)gridDim.x * kBlock) acc += r[i] * z[i];
  const double rr[1] = {block_sum(acc, lds)};
  grid_finish_fn<1>(rr, scratch, lds, [&](const double (&tot)[1]) {
    const double rz = tot[0];
    const double it = scal[SC_ITER] + 1.0;
    double done = 0.0;
    if (it == 0.0) {
      scal[SC_RZ0] = rz;
      scal[SC_BETA] = 0.0;
      if (!(rz > 0.0) || !isfinite(rz)) done = (rz == 0.0) ? 1.0 : 2.0;      // zero right-hand side: x = 0 is the solution
    } else {
      scal[SC_BETA] = rz / scal[SC_RZ];
      if (!isfinite(rz) || rz < 0.0) done = 2.0;
      else if (rz <= scal[SC_TOL2] * scal[SC_RZ0]) done = 1.0;
      else if (it >= scal[SC_MAXIT]) done = 3.0;
    }
    scal[SC_RZ] = rz;
    scal[SC_ITER] = it;
    scal[SC_DONE] = done;
    if (host4) {
      __hip_atomic_store(&host4[0], it, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __hip_atomic_store(&host4[1], done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __hip_atomic_store(&host4[2], rz, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __hip_atomic_store(&host4[3], scal[SC_RZ0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if (sig.seq_host) signal();
  });
}

__global__ __launch_bounds__(kBlock) void pcg_p_kernel(int n, double* __restrict__ p, const double* __restrict__ z, const double* scal) {
  if (scal[SC_DONE] != 0.0) return;
  const double beta = scal[SC_BETA];
  for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < n; i += (long long)gridDim.x * kBlock)
    p[i] = (beta == 0.0) ? z[i] : z[i] + beta * p[i];
}

YList make_ylist(const BarrierParams& P) {
  std::vector<unsigned short> rows[8];
  int base = 0;
  for (int ci = 0; ci < P.ncones; ++ci) {
    const ConeSpec& S = P.cone[ci];
    const int nact = S.nact();
    int slot = base;
    for (int a = 0; a < nact; ++a)
      for (int b = a; b < nact; ++b, ++slot) {
        const int ra = S.col(a), rb = S.col(b);
        rows[ra].push_back((unsigned short)(rb | slot << 8));
        if (ra != rb) rows[rb].push_back((unsigned short)(ra | slot << 8));
      }
    base += S.nY();
  }
  YList y{};
  int n = 0;
  for (int a = 0; a < 8; ++a) {
    y.ptr[a] = (unsigned char)n;
    for (unsigned short pq : rows[a]) y.pair[n++] = pq;
  }
  y.ptr[8] = (unsigned char)n;
  return y;
}

void launch_elop_phase1(hipStream_t st, const DevElOp& E, const BarrierParams& P, const double* Y, const double* v, const double* vmul,
                        const double* vscale, double* elbuf, int diag, const double* done) {
  const int epb = kBlock / E.tpe;
  const int npass = (E.nel + epb - 1) / epb;
  const size_t lds = (size_t)epb * E.slot_doubles * sizeof(double);
  const YList tab = make_ylist(P);
  // persistent workgroups: as many as are resident on the device at once (occupancy query: LDS and registers), each walking
  // its share of the passes with the next pass's loads in flight behind the current one's arithmetic
  static const int ncu = [] {
    int dev = 0, n = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
    return n > 0 ? n : 256;
  }();
#define MGB_ELOP(T)                                                                                                            \
  {                                                                                                                            \
    int per_cu = 0;                                                                                                            \
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, elop_apply_kernel<T>, kBlock, lds) != hipSuccess || per_cu < 1)  \
      per_cu = 2;                                                                                                              \
    const int grid = std::max(1, std::min(npass, ncu * per_cu));                                                               \
    hipLaunchKernelGGL(elop_apply_kernel<T>, dim3(grid), dim3(kBlock), lds, st, E, P.K, P.nY(), tab, Y, v, vmul, vscale, elbuf,  \
                       diag, done);                                                                                            \
  }
  switch (E.tpe) {
    case 8: MGB_ELOP(8); break;
    case 16: MGB_ELOP(16); break;
    case 32: MGB_ELOP(32); break;
    case 64: MGB_ELOP(64); break;
    case 128: MGB_ELOP(128); break;
    default: MGB_ELOP(256); break;
  }
#undef MGB_ELOP
}

}  // namespace

void launch_elop_assemble(hipStream_t st, const DevElOp& E, const DevElAsm& A, BarrierParams P, const double* Y, double* elmat,
                          double* avals) {
  if (!E.valid() || !A.valid()) return;
  const int epb = kBlock / E.tpe;
  const int npass = (E.nel + epb - 1) / epb;
  const int grid = std::max(1, std::min(npass, kMaxBlocks * 2));
  const size_t lds = (size_t)epb * A.slot_doubles * sizeof(double);
#define MGB_ELASM(T) hipLaunchKernelGGL(elop_assemble_kernel<T>, dim3(grid), dim3(kBlock), lds, st, E, A, P.nY(), Y, elmat)
  switch (E.tpe) {
    case 8: MGB_ELASM(8); break;
    case 16: MGB_ELASM(16); break;
    case 32: MGB_ELASM(32); break;
    case 64: MGB_ELASM(64); break;
    case 128: MGB_ELASM(128); break;
    default: MGB_ELASM(256); break;
  }
#undef MGB_ELASM
  hipLaunchKernelGGL(gather_sum_kernel, dim3(grid_for(A.nnzA)), dim3(kBlock), 0, st, A.nnzA, A.aptr, A.aidx, elmat, avals);
}

void launch_elop_apply(hipStream_t st, const DevElOp& E, BarrierParams P, const double* Y, double* elbuf, const MgEpi& epi) {
  if (!E.valid() || E.N == 0) return;
  // MG_FIRST forms its input c0 Dinv b on the fly
  const double* vmul = epi.mode == MG_FIRST ? epi.dinv : epi.vmul;
  const double* vscale = epi.mode == MG_FIRST ? epi.coef : epi.vscale;
  const double* v = epi.mode == MG_FIRST ? epi.b : epi.v;
  launch_elop_phase1(st, E, P, Y, v, vmul, vscale, elbuf, 0, epi.done);
  hipLaunchKernelGGL(dof_gather_kernel, dim3(grid_for(E.N)), dim3(kBlock), 0, st, E, elbuf, epi);
}

void launch_elop_diaginv(hipStream_t st, const DevElOp& E, BarrierParams P, const double* Y, double* elbuf, double* dinv) {
  if (!E.valid() || E.N == 0) return;
  launch_elop_phase1(st, E, P, Y, nullptr, nullptr, nullptr, elbuf, 1, nullptr);
  MgEpi epi;
  epi.mode = MG_DIAGINV;
  epi.n = E.N;
  epi.out = dinv;
  hipLaunchKernelGGL(dof_gather_kernel, dim3(grid_for(E.N)), dim3(kBlock), 0, st, E, elbuf, epi);
}

void launch_csr_apply(hipStream_t st, const DevCsr& A, const MgEpi& epi_in) {
  if (A.rows == 0) return;
  MgEpi epi = epi_in;
  if (epi.mode == MG_FIRST) {      // input c0 Dinv b on the fly: the scale is applied to the row sum (linear)
    epi.v = epi.b;
    epi.vmul = epi.dinv;
    epi.vscale = epi.coef;
  }
#define MGB_CSR_APPLY(G)                                                                                                       \
  hipLaunchKernelGGL(csr_apply_kernel<G>, dim3(grid_for((long long)A.rows * G)), dim3(kBlock), 0, st, A.rows, A.rowptr, A.colidx, \
                     A.vals, epi)
  switch (A.group) {
    case 1: MGB_CSR_APPLY(1); break;
    case 2: MGB_CSR_APPLY(2); break;
    case 4: MGB_CSR_APPLY(4); break;
    case 8: MGB_CSR_APPLY(8); break;
    case 16: MGB_CSR_APPLY(16); break;
    case 32: MGB_CSR_APPLY(32); break;
    default: MGB_CSR_APPLY(64); break;
  }
#undef MGB_CSR_APPLY
}

void launch_expand_sym(hipStream_t st, int nnz_full, const int* map, const double* lower, double* full, int N, const int* diagpos,
                       double* dinv) {
  if (nnz_full == 0) return;
  hipLaunchKernelGGL(expand_sym_kernel, dim3(grid_for(nnz_full)), dim3(kBlock), 0, st, nnz_full, map, lower, full, N, diagpos, dinv);
}

void launch_cheb_coef(hipStream_t st, const double* scal, double* coef, int degree, double lo_frac, double hi_frac) {
  hipLaunchKernelGGL(cheb_coef_kernel, dim3(1), dim3(64), 0, st, scal, coef, degree, lo_frac, hi_frac);
}

void mg_device_init() {
  (void)hipFuncSetAttribute((const void*)dense_inverse_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                            kDenseMax * (kDenseMax + 1) * (int)sizeof(double));
}

void launch_dense_inverse(hipStream_t st, int N, const int* rowptr, const int* colidx, const double* vals, double* Ainv, int* fail) {
  if (N == 0) return;
  hipLaunchKernelGGL(dense_inverse_kernel, dim3(1), dim3(kBlock), (size_t)N * (N + 1) * sizeof(double), st, N, rowptr, colidx, vals,
                     Ainv, fail);
}

void launch_dense_apply(hipStream_t st, int N, const double* Ainv, const double* b, double* x, const double* done) {
  if (N == 0) return;
  hipLaunchKernelGGL(dense_apply_kernel, dim3((N + kBlock / 64 - 1) / (kBlock / 64)), dim3(kBlock), 0, st, N, Ainv, b, x, done);
}

void launch_pcg_init(hipStream_t st, int n, const double* g, double* x, double* r, double* scal, double rtol, int maxit) {
  hipLaunchKernelGGL(pcg_init_kernel, dim3(grid_for(n)), dim3(kBlock), 0, st, n, g, x, r, scal, rtol, maxit);
}

void launch_pcg_update(hipStream_t st, int n, double* x, double* r, const double* p, const double* Ap, const double* scal) {
  hipLaunchKernelGGL(pcg_update_kernel, dim3(grid_for(n)), dim3(kBlock), 0, st, n, x, r, p, Ap, scal);
}

void launch_pcg_dot(hipStream_t st, int n, const double* r, const double* z, double* scal, double* scratch, double* host4,
                    HostSignal sig) {
  hipLaunchKernelGGL(pcg_dot_kernel, dim3(grid_for(n)), dim3(kBlock), 0, st, n, r, z, scal, scratch, host4, sig);
}

void launch_pcg_p(hipStream_t st, int n, double* p, const double* z, const double* scal) {
  hipLaunchKernelGGL(pcg_p_kernel, dim3(grid_for(n)), dim3(kBlock), 0, st, n, p, z, scal);
}

void launch_power_start(hipStream_t st, int n, const double* ev, const double* dinv, double* scal, double* scratch) {
  hipLaunchKernelGGL(power_start_kernel, dim3(grid_for(n)), dim3(kBlock), 0, st, n, ev, dinv, scal, scratch);
}

}  // namespace mgb
