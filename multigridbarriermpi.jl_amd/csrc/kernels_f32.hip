// Float32 instantiation of the bandwidth-shaped kernels of the Newton path (the reference runs
// Float32 on its Metal backend, test/test_utils.jl:67-88): row-block CSR SpMV (apply_D, restriction, Hessian assembly) and the
// barrier kernels f0 / f1 / f2 of the power-cone / half-space family.
//
// The kernels are the production kernels of kernels.hip written once more as templates over the scalar type: same lane
// groups, same fixed-order shuffle tree, same expressions per row.  The T = double instantiation exists only for the tests,
// which check it BIT FOR BIT against the production kernels -- so the float instantiation is the same algorithm with half
// the bytes per entry (4-byte values and vectors; the index arrays are shared with the double-precision operators).
// There is no Float32 factorisation: on MI355X fp64 runs at the fp32 vector rate and the direct solve is latency-bound, so
// the solver stays in double; these entry points cover the operator / barrier evaluation (mgb_amg_f0_f32 / f1 / f2).
#include "kernels.hpp"

#include <algorithm>

namespace mgb {

namespace {

constexpr int kBlock = 256;
constexpr int kMaxBlocks = 2048;
constexpr int kMaxK = 8;

inline int grid_for(long long work_items) {
  long long b = (work_items + kBlock - 1) / kBlock;
  return (int)std::min<long long>(std::max<long long>(b, 1), kMaxBlocks);
}

__device__ inline unsigned xcd_block(unsigned b, unsigned nb) {      // kernels.hip: XCD-aware block order
  if (nb < 16u) return b;
  const unsigned per = nb >> 3, main = per << 3;
  return b < main ? (b & 7u) * per + (b >> 3) : b;
}

constexpr int kSpmvU = 2;
template <int G, class T>
__global__ __launch_bounds__(kBlock) void spmv_kernel_t(int rows, const int* __restrict__ rowptr, const int* __restrict__ colidx,
                                                         const T* __restrict__ vals, const T* __restrict__ x, const T* y0, T* y) {
  const int lane = threadIdx.x % G;
  const long long stride = (long long)gridDim.x * (kBlock / G);
  for (long long row0 = (long long)xcd_block(blockIdx.x, gridDim.x) * (kBlock / G) + threadIdx.x / G; row0 < rows;
       row0 += kSpmvU * stride) {
    int b[kSpmvU], e[kSpmvU];
    T acc[kSpmvU], base[kSpmvU];
#pragma unroll
    for (int u = 0; u < kSpmvU; ++u) {
      const long long row = row0 + u * stride;
      const bool ok = row < rows;
      b[u] = ok ? rowptr[row] : 0;
      e[u] = ok ? rowptr[row + 1] : 0;
      base[u] = (ok && y0 && lane == 0) ? y0[row] : T(0);
      acc[u] = T(0);
    }
    int ci[kSpmvU];
    T va[kSpmvU];
#pragma unroll
    for (int u = 0; u < kSpmvU; ++u) {
      const int k = b[u] + lane;
      const bool in = k < e[u];
      ci[u] = in ? colidx[k] : -1;
      va[u] = in ? vals[k] : T(0);
    }
#pragma unroll
    for (int u = 0; u < kSpmvU; ++u) acc[u] = (ci[u] >= 0) ? va[u] * x[ci[u]] : T(0);
#pragma unroll
    for (int u = 0; u < kSpmvU; ++u)
      for (int k = b[u] + lane + G; k < e[u]; k += G) acc[u] += vals[k] * x[colidx[k]];
#pragma unroll
    for (int u = 0; u < kSpmvU; ++u) {
      T a = acc[u];
#pragma unroll
      for (int o = G / 2; o > 0; o >>= 1) a += __shfl_down(a, o, G);
      const long long row = row0 + u * stride;
      if (lane == 0 && row < rows) y[row] = base[u] + a;
    }
  }
}

template <class T>
struct ConeT {
  T q[3];
  T s, phi, sa;
  bool ok;
};

template <class T>
__device__ inline T pow_a(T s, T a) {
  if (a == T(2)) return s * s;
  if (a == T(1)) return s;
  return pow(s, a);
}

template <class T>
__device__ inline ConeT<T> load_cone(const ConeSpec& P, const T* dz) {
  ConeT<T> c;
  if (P.kind == 1) {
    T phi = T(P.off);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      c.q[i] = (i < P.nq) ? dz[P.iq[i]] : T(0);
      phi += (i < P.nq) ? T(P.coef[i]) * c.q[i] : T(0);
    }
    c.s = T(1);
    c.sa = T(1);
    c.phi = phi;
    c.ok = phi > T(0);
    return c;
  }
  T qq = T(0);
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    c.q[i] = (i < P.nq) ? dz[P.iq[i]] : T(0);
    qq += c.q[i] * c.q[i];
  }
  c.s = dz[P.is] + (P.is2 >= 0 ? dz[P.is2] : T(0));
  c.ok = c.s > T(0);
  c.sa = c.ok ? pow_a<T>(c.s, T(P.a)) : T(-1);
  c.phi = c.sa - qq;
  c.ok = c.ok && (c.phi > T(0));
  return c;
}

template <class T>
__device__ inline T pick3(const T (&q)[3], int i) {
  return i == 0 ? q[0] : (i == 1 ? q[1] : q[2]);
}

// per row: w F(Dz) and w <c, Dz> (double outputs: the sums over the rows are formed in double by the caller's reduction)
template <class T>
__global__ __launch_bounds__(kBlock) void barrier_f0_rows_kernel_t(int n, BarrierParams P, const T* __restrict__ Dz,
                                                                    const T* __restrict__ w, const T* __restrict__ c,
                                                                    double* __restrict__ outF, double* __restrict__ outC) {
  for (long long q = (long long)blockIdx.x * kBlock + threadIdx.x; q < n; q += (long long)gridDim.x * kBlock) {
    const T* dz = Dz + q * P.K;
    T F = T(0);
    for (int ci = 0; ci < P.ncones; ++ci) {
      const ConeT<T> k = load_cone<T>(P.cone[ci], dz);
      F += k.ok ? (-log(k.phi) - T(P.cone[ci].mu) * log(k.s)) : T(INFINITY);
    }
    T cd = T(0);
#pragma unroll
    for (int j = 0; j < kMaxK; ++j) cd += (j < P.K) ? c[q * P.K + j] * dz[j] : T(0);
    outF[q] = (double)(w[q] * F);
    outC[q] = (double)(w[q] * cd);
  }
}

template <class T>
__global__ __launch_bounds__(kBlock) void barrier_f1_kernel_t(int n, BarrierParams P, const T* __restrict__ Dz,
                                                               const T* __restrict__ w, const T* __restrict__ c, T t,
                                                               T* __restrict__ v) {
  for (long long q = (long long)blockIdx.x * kBlock + threadIdx.x; q < n; q += (long long)gridDim.x * kBlock) {
    const T* dz = Dz + q * P.K;
    const T* cq = c + q * P.K;
    const T wq = w[q];
    T vr[kMaxK];
#pragma unroll
    for (int j = 0; j < kMaxK; ++j) vr[j] = (j < P.K) ? wq * (t * cq[j]) : T(0);
    for (int ci = 0; ci < P.ncones; ++ci) {
      const ConeSpec& S = P.cone[ci];
      const ConeT<T> k = load_cone<T>(S, dz);
      if (S.kind == 1) {
#pragma unroll
        for (int j = 0; j < kMaxK; ++j) {
          T add = T(0);
#pragma unroll
          for (int i = 0; i < 3; ++i) add += (i < S.nq && S.iq[i] == j) ? -wq * (T(S.coef[i]) / k.phi) : T(0);
          vr[j] += add;
        }
        continue;
      }
      const T ds = T(S.a) * pow_a<T>(k.s, T(S.a) - T(1));
      const T gs = wq * (-ds / k.phi - T(S.mu) / k.s);
#pragma unroll
      for (int j = 0; j < kMaxK; ++j) {
        T add = T(0);
#pragma unroll
        for (int i = 0; i < 3; ++i) add += (i < S.nq && S.iq[i] == j) ? wq * (T(2) * k.q[i] / k.phi) : T(0);
        add += (S.is == j) ? gs : T(0);
        add += (S.is2 == j) ? gs : T(0);
        vr[j] += add;
      }
    }
    T* vq = v + q * P.K;
#pragma unroll
    for (int j = 0; j < kMaxK; ++j)
      if (j < P.K) vq[j] = vr[j];
  }
}

template <class T>
__global__ __launch_bounds__(kBlock) void barrier_f2_kernel_t(int n, BarrierParams P, const T* __restrict__ Dz,
                                                               const T* __restrict__ w, T* __restrict__ Y) {
  const int nY = P.nY();
  for (long long q = (long long)blockIdx.x * kBlock + threadIdx.x; q < n; q += (long long)gridDim.x * kBlock) {
    const T* dz = Dz + q * P.K;
    T* yq = Y + q * nY;
    const T wq = w[q];
    int slot = 0;
    for (int ci = 0; ci < P.ncones; ++ci) {
      const ConeSpec& S = P.cone[ci];
      const ConeT<T> k = load_cone<T>(S, dz);
      if (S.kind == 1) {
        const T ip2l = T(1) / (k.phi * k.phi);
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
          for (int j = 0; j < 3; ++j)
            if (i < S.nq && j >= i && j < S.nq) yq[slot++] = wq * (T(S.coef[i]) * T(S.coef[j]) * ip2l);
        continue;
      }
      const T a = T(S.a);
      const T ds = a * pow_a<T>(k.s, a - T(1));
      const T dds = (a == T(1)) ? T(0) : a * (a - T(1)) * pow_a<T>(k.s, a - T(2));
      const T ip = T(1) / k.phi, ip2 = ip * ip;
      const T hss = -dds * ip + ds * ds * ip2 + T(S.mu) / (k.s * k.s);
      const int nq = S.nq, nact = S.nact();
#pragma unroll
      for (int i = 0; i < 5; ++i)
#pragma unroll
        for (int j = 0; j < 5; ++j) {
          if (i < nact && j >= i && j < nact) {
            const int ai = min(i, nq), aj = min(j, nq);
            T h;
            if (aj < nq) h = T(4) * pick3<T>(k.q, ai) * pick3<T>(k.q, aj) * ip2 + (ai == aj ? T(2) * ip : T(0));
            else if (ai < nq) h = T(-2) * pick3<T>(k.q, ai) * ds * ip2;
            else h = hss;
            yq[slot++] = wq * h;
          }
        }
    }
  }
}

template <class A, class B>
__global__ __launch_bounds__(kBlock) void convert_kernel(long long n, const A* __restrict__ in, B* __restrict__ out) {
  for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < n; i += (long long)gridDim.x * kBlock) out[i] = (B)in[i];
}

template <int G, class T>
void spmv_launch_t(hipStream_t st, const DevCsr& A, const T* vals, const T* x, const T* y0, T* y) {
  hipLaunchKernelGGL((spmv_kernel_t<G, T>), dim3(grid_for((long long)A.rows * G)), dim3(kBlock), 0, st, A.rows, A.rowptr, A.colidx,
                     vals, x, y0, y);
}

template <class T>
void spmv_t(hipStream_t st, const DevCsr& A, const T* vals, const T* x, const T* y0, T* y) {
  if (A.rows == 0) return;
  switch (A.group) {
    case 1: spmv_launch_t<1, T>(st, A, vals, x, y0, y); break;
    case 2: spmv_launch_t<2, T>(st, A, vals, x, y0, y); break;
    case 4: spmv_launch_t<4, T>(st, A, vals, x, y0, y); break;
    case 8: spmv_launch_t<8, T>(st, A, vals, x, y0, y); break;
    case 16: spmv_launch_t<16, T>(st, A, vals, x, y0, y); break;
    case 32: spmv_launch_t<32, T>(st, A, vals, x, y0, y); break;
    default: spmv_launch_t<64, T>(st, A, vals, x, y0, y); break;
  }
}

}  // namespace

void launch_spmv_f32(hipStream_t st, const DevCsr& A, const float* vals, const float* x, const float* y0, float* y) {
  spmv_t<float>(st, A, vals, x, y0, y);
}
void launch_barrier_f0_rows_f32(hipStream_t st, int n, BarrierParams P, const float* Dz, const float* w, const float* c,
                                double* outF, double* outC) {
  if (n > 0) hipLaunchKernelGGL(barrier_f0_rows_kernel_t<float>, dim3(grid_for(n)), dim3(kBlock), 0, st, n, P, Dz, w, c, outF, outC);
}
void launch_barrier_f1_f32(hipStream_t st, int n, BarrierParams P, const float* Dz, const float* w, const float* c, float t, float* v) {
  if (n > 0) hipLaunchKernelGGL(barrier_f1_kernel_t<float>, dim3(grid_for(n)), dim3(kBlock), 0, st, n, P, Dz, w, c, t, v);
}
void launch_barrier_f2_f32(hipStream_t st, int n, BarrierParams P, const float* Dz, const float* w, float* Y) {
  if (n > 0) hipLaunchKernelGGL(barrier_f2_kernel_t<float>, dim3(grid_for(n)), dim3(kBlock), 0, st, n, P, Dz, w, Y);
}
void launch_to_f32(hipStream_t st, long long n, const double* in, float* out) {
  if (n > 0) hipLaunchKernelGGL((convert_kernel<double, float>), dim3(grid_for(n)), dim3(kBlock), 0, st, n, in, out);
}

// T = double instantiation of the same templates (tests only: must reproduce the production kernels bit for bit)
void launch_spmv_tpl_f64(hipStream_t st, const DevCsr& A, const double* x, const double* y0, double* y) {
  spmv_t<double>(st, A, A.vals, x, y0, y);
}
void launch_barrier_f1_tpl_f64(hipStream_t st, int n, BarrierParams P, const double* Dz, const double* w, const double* c, double t,
                               double* v) {
  if (n > 0) hipLaunchKernelGGL(barrier_f1_kernel_t<double>, dim3(grid_for(n)), dim3(kBlock), 0, st, n, P, Dz, w, c, t, v);
}
void launch_barrier_f2_tpl_f64(hipStream_t st, int n, BarrierParams P, const double* Dz, const double* w, double* Y) {
  if (n > 0) hipLaunchKernelGGL(barrier_f2_kernel_t<double>, dim3(grid_for(n)), dim3(kBlock), 0, st, n, P, Dz, w, Y);
}

}  // namespace mgb
