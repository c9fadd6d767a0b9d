// Micro-timings of the single-wave building blocks of the factorisation chain, in isolation: the 32x32 pivot-block factor
// (factor_diag_block) and the panel substitution with one, two and three row blocks per wave (trsm_quad_n), each run
// `reps` times back to back by ONE workgroup and timed with the 100 MHz wall clock.  The chain's panel step is bound by
// exactly these two (tools/chol_timeline.py stamps), so this is the quick loop for changing them.
//   build:  hipcc --offload-arch=gfx950 -O3 -std=c++17 -I multigridbarriermpi.jl_amd/csrc -I include tools/chol_micro.hip \
//             -L multigridbarriermpi.jl_amd/lib -lmgb_hip -Wl,-rpath,'$ORIGIN/../../multigridbarriermpi.jl_amd/lib' -o tools/_bin/chol_micro
//   run:    tools/_bin/chol_micro [reps]
#include "../multigridbarriermpi.jl_amd/csrc/gpuchol.hip"

#include <vector>

namespace mgb {
namespace {

__global__ __launch_bounds__(TB) void factor_micro(const double* Din, double* Lout, long long* ticks, int* fail, int reps) {
  __shared__ double D[PB * LP];
  __shared__ double Lo[PB * LP];
  for (int idx = threadIdx.x; idx < PB * PB; idx += TB) D[(idx / PB) * LP + idx % PB] = Din[idx];
  __syncthreads();
  const long long t0 = wall_clock64();
  for (int r = 0; r < reps; ++r) {
    asm volatile("" ::: "memory");
    factor_diag_block(D, PB, Lo, nullptr, fail, nullptr);
    __syncthreads();
  }
  const long long t1 = wall_clock64();
  if (threadIdx.x == 0) ticks[0] = t1 - t0;
  for (int idx = threadIdx.x; idx < PB * PB; idx += TB) Lout[idx] = Lo[(idx / PB) * LP + idx % PB];
}

// NB row blocks per wave against a factor in quad order; the staged rows are re-read every repetition
template <int NB>
__global__ __launch_bounds__(TB) void trsm_micro(const double* Lin, const double* Ain, double* Xout, long long* ticks, int reps) {
  constexpr int TP = TS + 8;
  __shared__ double Lq[PB * PB];
  __shared__ double AT[3][PB * TP];
  for (int idx = threadIdx.x; idx < PB * PB; idx += TB) Lq[lq_index(idx / PB, idx % PB)] = Lin[idx];      // Lin row-major L[m][j]
  for (int b = 0; b < 3; ++b)
    for (int idx = threadIdx.x; idx < PB * TS; idx += TB) AT[b][(idx / TS) * TP + idx % TS] = Ain[b * PB * TS + idx];
  __syncthreads();
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, r = 16 * w + (lane >> 2), c4 = lane & 3;
  double f[NB][PB / 4];
  const long long t0 = wall_clock64();
  for (int rep = 0; rep < reps; ++rep) {
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int k = 0; k < PB / 4; ++k) {
        f[b][k] = AT[b][(c4 + 4 * k) * TP + r];
        asm volatile("" : "+v"(f[b][k]));      // opaque: the repetitions are not loop-invariant to the compiler
      }
    trsm_quad_n<NB>(f, Lq, c4);
    __syncthreads();
  }
  const long long t1 = wall_clock64();
  if (threadIdx.x == 0) ticks[0] = t1 - t0;
#pragma unroll
  for (int b = 0; b < NB; ++b)
#pragma unroll
    for (int k = 0; k < PB / 4; ++k) Xout[(b * TS + r) * PB + c4 + 4 * k] = f[b][k];
}

}  // namespace
}  // namespace mgb

#define CK(x) do { hipError_t err_ = (x); if (err_ != hipSuccess) { std::printf("HIP error %s at line %d\n", hipGetErrorString(err_), __LINE__); return 2; } } while (0)

int main(int argc, char** argv) {
  using namespace mgb;
  const int reps = argc > 1 ? std::atoi(argv[1]) : 200;
  std::vector<double> M(PB * PB), A(PB * PB, 0.0), L(PB * PB, 0.0), B(3 * PB * TS);
  unsigned s = 12345u;
  auto rnd = [&] { s = s * 1664525u + 1013904223u; return (double)(s >> 8) / (1 << 24) - 0.5; };
  for (auto& v : M) v = rnd();
  for (int i = 0; i < PB; ++i)
    for (int j = 0; j <= i; ++j) {
      double a = (i == j) ? 8.0 : 0.0;
      for (int k = 0; k < PB; ++k) a += M[i * PB + k] * M[j * PB + k];
      A[i * PB + j] = a;
    }
  for (auto& v : B) v = rnd();
  // host Cholesky (row-major lower L)
  for (int j = 0; j < PB; ++j) {
    double d = A[j * PB + j];
    for (int k = 0; k < j; ++k) d -= L[j * PB + k] * L[j * PB + k];
    L[j * PB + j] = std::sqrt(d);
    for (int i = j + 1; i < PB; ++i) {
      double v = A[i * PB + j];
      for (int k = 0; k < j; ++k) v -= L[i * PB + k] * L[j * PB + k];
      L[i * PB + j] = v / L[j * PB + j];
    }
  }
  double *dA, *dL, *dB, *dX;
  long long* dt;
  int* dfail;
  CK(hipMalloc(&dA, PB * PB * 8));
  CK(hipMalloc(&dL, PB * PB * 8));
  CK(hipMalloc(&dB, 3 * PB * TS * 8));
  CK(hipMalloc(&dX, 3 * TS * PB * 8));
  CK(hipMalloc(&dt, 8));
  CK(hipMalloc(&dfail, 4));
  CK(hipMemset(dfail, 0, 4));
  CK(hipMemcpy(dA, A.data(), PB * PB * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(dB, B.data(), 3 * PB * TS * 8, hipMemcpyHostToDevice));
  long long t = 0;
  std::vector<double> Ld(PB * PB);
  for (int pass = 0; pass < 2; ++pass) {      // first pass warms the instruction cache
    hipLaunchKernelGGL(factor_micro, dim3(1), dim3(TB), 0, 0, dA, dL, dt, dfail, reps);
    CK(hipDeviceSynchronize());
  }
  CK(hipMemcpy(&t, dt, 8, hipMemcpyDeviceToHost));
  CK(hipMemcpy(Ld.data(), dL, PB * PB * 8, hipMemcpyDeviceToHost));
  double err = 0;
  for (int i = 0; i < PB; ++i)
    for (int j = 0; j <= i; ++j) err = std::max(err, std::fabs(Ld[i * PB + j] - (i == j ? 1.0 / L[i * PB + i] : L[i * PB + j])));
  std::printf("factor 32x32: %.0f ns per block (%d reps), max |L - host| %.2e\n", 10.0 * t / reps, reps, err);
  // the device factor (reciprocal diagonal) feeds the substitutions
  std::vector<double> X(3 * TS * PB);
  auto check = [&](int nb) {
    double e = 0;
    for (int b = 0; b < nb; ++b)
      for (int r = 0; r < TS; ++r) {      // X L' = A  =>  sum_j X[r][j] L[m][j] = A[r][m]
        for (int m = 0; m < PB; ++m) {
          double acc = 0;
          for (int j = 0; j <= m; ++j) acc += X[(b * TS + r) * PB + j] * L[m * PB + j];
          e = std::max(e, std::fabs(acc - B[b * PB * TS + m * TS + r]));
        }
      }
    return e;
  };
#define RUN(NB)                                                                                              \
  do {                                                                                                       \
    for (int pass = 0; pass < 2; ++pass) {                                                                   \
      hipLaunchKernelGGL(trsm_micro<NB>, dim3(1), dim3(TB), 0, 0, dL, dB, dX, dt, reps);                     \
      CK(hipDeviceSynchronize());                                                                            \
    }                                                                                                        \
    CK(hipMemcpy(&t, dt, 8, hipMemcpyDeviceToHost));                                                         \
    CK(hipMemcpy(X.data(), dX, X.size() * 8, hipMemcpyDeviceToHost));                                        \
    std::printf("substitution, %d block(s) of 16 rows per wave: %.0f ns (%d reps), residual %.2e\n", NB, 10.0 * t / reps, reps, check(NB)); \
  } while (0)
  RUN(1);
  RUN(2);
  RUN(3);
  return 0;
}
