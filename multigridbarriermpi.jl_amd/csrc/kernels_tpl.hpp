// The bandwidth-shaped kernels of the Newton path written ONCE, as templates over the scalar type: row-block CSR SpMV (apply_D,
// restriction, Hessian assembly: a3, a5, a6 of SURVEY.md section 8) and the per-row barrier evaluations F1 / F2 (map_rows of the barrier,
// src/MultiGridBarrierMPI.jl:161-170).  kernels.hip instantiates them for double -- these ARE the production kernels -- and
// kernels_f32.hip for float (the reference runs Float32 on its Metal backend, test/test_utils.jl:67-88).
// HIP-only header.
#pragma once
#include "devutil.hpp"

namespace mgb {
namespace {

constexpr int kMaxK = 8;      // rows of D (capi.cpp rejects larger problems for the barrier kernels)

// G lanes cooperate on one row: lane j reads nonzero j, j+G, ... (coalesced across the group and, because consecutive rows are
// adjacent in CSR storage, across the 64/G rows of a wave); the G partial sums are combined with a fixed-order shuffle tree
// -> bitwise reproducible.  kSpmvU rows are in flight per lane group (memory-level parallelism of the dependent
// rowptr -> nonzero -> x chain).
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

// exponent data of term ci at node q: the per-node arrays of an x-dependent exponent, or the term's constants
template <class T>
struct ConeAM {
  T a, mu;
};
template <class T>
__device__ inline ConeAM<T> cone_am(const BarrierParams& P, int ci, long long q) {
  ConeAM<T> r;
  r.a = P.a_node ? T(P.a_node[q * P.ncones + ci]) : T(P.cone[ci].a);
  r.mu = P.mu_node ? T(P.mu_node[q * P.ncones + ci]) : T(P.cone[ci].mu);
  return r;
}

template <class T>
__device__ inline ConeT<T> load_cone(const ConeSpec& P, const T* dz, T a) {
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
  c.sa = c.ok ? pow_a<T>(c.s, a) : T(-1);
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
      if (!P.active(ci, q)) continue;
      const ConeAM<T> am = cone_am<T>(P, ci, q);
      const ConeT<T> k = load_cone<T>(P.cone[ci], dz, am.a);
      F += k.ok ? (-log(k.phi) - am.mu * log(k.s)) : T(INFINITY);
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
      if (!P.active(ci, q)) continue;
      const ConeAM<T> am = cone_am<T>(P, ci, q);
      const ConeT<T> k = load_cone<T>(S, dz, am.a);
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
      const T ds = am.a * pow_a<T>(k.s, am.a - T(1));
      const T gs = wq * (-ds / k.phi - am.mu / k.s);
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
      if (!P.active(ci, q)) {      // inactive piece: its slots are zeros
        const int ns = S.nY();
        for (int i = 0; i < ns; ++i) yq[slot++] = T(0);
        continue;
      }
      const ConeAM<T> am = cone_am<T>(P, ci, q);
      const ConeT<T> k = load_cone<T>(S, dz, am.a);
      if (S.kind == 1) {
        const T ip2l = T(1) / (k.phi * k.phi);
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
          for (int j = 0; j < 3; ++j)
            if (i < S.nq && j >= i && j < S.nq) yq[slot++] = wq * (T(S.coef[i]) * T(S.coef[j]) * ip2l);
        continue;
      }
      const T a = am.a;
      const T ds = a * pow_a<T>(k.s, a - T(1));
      const T dds = (a == T(1)) ? T(0) : a * (a - T(1)) * pow_a<T>(k.s, a - T(2));
      const T ip = T(1) / k.phi, ip2 = ip * ip;
      const T hss = -dds * ip + ds * ds * ip2 + am.mu / (k.s * k.s);
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


}  // namespace
}  // namespace mgb
