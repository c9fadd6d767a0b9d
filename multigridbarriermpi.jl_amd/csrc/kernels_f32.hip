// Float32 instantiation of the bandwidth-shaped kernels of the Newton path (the reference runs
// Float32 on its Metal backend, test/test_utils.jl:67-88): row-block CSR SpMV (apply_D, restriction, Hessian assembly) and the
// barrier kernels f0 / f1 / f2 of the power-cone / half-space family.
//
// The kernels are the templates of kernels_tpl.hpp, whose T = double instantiation is the production code (kernels.hip): the
// float instantiation is the same algorithm with half the bytes per entry (4-byte values and vectors; the index arrays are
// shared with the double-precision operators).  The *_tpl_f64 launches below run the double instantiation from THIS translation
// unit; the tests check it bit for bit against the production entry points.
// There is no Float32 factorisation: on MI355X fp64 runs at the fp32 vector rate and the direct solve is latency-bound, so
// the solver stays in double; these entry points cover the operator / barrier evaluation (mgb_amg_f0_f32 / f1 / f2).
#include "kernels.hpp"

#include <algorithm>

#include "kernels_tpl.hpp"

namespace mgb {

namespace {

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
