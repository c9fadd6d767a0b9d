// Latency / issue cost of the instructions the single-wave parts of the factorisation chain are made of, measured on ONE wave
// (and on four, one per SIMD) with the shader clock (s_memtime) beside the 100 MHz wall clock: dependent and independent
// v_fma_f64, v_rsq_f64 + refinement, v_readlane -> VALU, LDS write -> read, DPP quad broadcast, v_permlane32_swap.
//   build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/isa_micro.hip -o tools/_bin/isa_micro
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t err_ = (x); if (err_ != hipSuccess) { std::printf("HIP error %s at line %d\n", hipGetErrorString(err_), __LINE__); return 2; } } while (0)

constexpr int N = 512;      // operations per timed region

template <int OP>
__global__ __launch_bounds__(256) void micro(double* out, long long* ticks, double seed) {
  __shared__ double lds[512];
  const int lane = threadIdx.x & 63;
  double x = seed + lane * 1e-3, y = seed * 0.5, a = 1.0000001, b = 1e-9;
  double acc[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) acc[q] = seed + q;
  lds[threadIdx.x] = x;
  lds[256 + threadIdx.x] = y;
  __syncthreads();
  asm volatile("" : "+v"(x), "+v"(y));
#pragma unroll
  for (int q = 0; q < 8; ++q) asm volatile("" : "+v"(acc[q]));
  const long long w0 = wall_clock64();
  const long long c0 = clock64();
  asm volatile("" : "+v"(x), "+v"(y));
#pragma unroll
  for (int q = 0; q < 8; ++q) asm volatile("" : "+v"(acc[q]));
  if (OP == 0) {      // dependent v_fma_f64
#pragma unroll
    for (int k = 0; k < N; ++k) x = fma(x, a, b);
  } else if (OP == 1) {      // eight independent chains
#pragma unroll
    for (int k = 0; k < N / 8; ++k)
#pragma unroll
      for (int q = 0; q < 8; ++q) acc[q] = fma(acc[q], a, b);
  } else if (OP == 2) {      // v_rsq_f64 + the correction (5 more dependent operations)
#pragma unroll
    for (int k = 0; k < N / 8; ++k) {
      const double r = __builtin_amdgcn_rsq(x);
      const double e = fma(r * -x, r, 1.0);
      x = fma(r * e, fma(e, 0.375, 0.5), r) + 1.5;
    }
  } else if (OP == 3) {      // v_readlane (two halves) -> VALU that reads the SGPRs
#pragma unroll
    for (int k = 0; k < N / 4; ++k) {
      const int lo = __builtin_amdgcn_readlane(__double2loint(x), 7), hi = __builtin_amdgcn_readlane(__double2hiint(x), 7);
      x = x + __hiloint2double(hi, lo);
    }
  } else if (OP == 4) {      // LDS write -> read of another lane's word -> write
#pragma unroll
    for (int k = 0; k < N / 8; ++k) {
      lds[lane] = x;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      x = lds[lane ^ 1] + 1.0;
    }
  } else if (OP == 5) {      // DPP quad broadcast of a double -> fma
#pragma unroll
    for (int k = 0; k < N / 4; ++k) {
      const int lo = __builtin_amdgcn_mov_dpp(__double2loint(x), 0x55, 0xf, 0xf, false);
      const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(x), 0x55, 0xf, 0xf, false);
      x = fma(__hiloint2double(hi, lo), a, b);
    }
  } else if (OP == 6) {      // v_permlane32_swap of a double -> fma
#pragma unroll
    for (int k = 0; k < N / 4; ++k) {
      const int lo = __double2loint(x), hi = __double2hiint(x);
      x = fma(__hiloint2double(__builtin_amdgcn_permlane32_swap(hi, hi, false, false)[0], __builtin_amdgcn_permlane32_swap(lo, lo, false, false)[0]), a, b);
    }
  } else if (OP == 7) {      // independent LDS b128 reads (issue cost), consumed at the end
    typedef double d2 __attribute__((ext_vector_type(2)));
    const d2* p = (const d2*)lds;
    d2 s = {0.0, 0.0};
#pragma unroll
    for (int k = 0; k < N / 4; ++k) {
      const d2 v = p[(lane + k) & 127];
      s += v;
    }
    x = s[0] + s[1];
  } else if (OP == 8) {      // dependent v_mul_f64
#pragma unroll
    for (int k = 0; k < N; ++k) x = x * a;
  } else if (OP == 10) {      // two interleaved dependent v_fma_f64 chains
#pragma unroll
    for (int k = 0; k < N / 2; ++k) {
      x = fma(x, a, b);
      y = fma(y, a, b);
    }
    x += y;
  } else if (OP == 11) {      // four interleaved dependent chains
#pragma unroll
    for (int k = 0; k < N / 4; ++k)
#pragma unroll
      for (int q = 0; q < 4; ++q) acc[q] = fma(acc[q], a, b);
  } else if (OP == 12) {      // independent ds_read_b128, no address arithmetic (constant offsets), summed pairwise at the end
    typedef double d2 __attribute__((ext_vector_type(2)));
    const d2* p = (const d2*)lds + lane;
    d2 v[16];
#pragma unroll
    for (int r = 0; r < N / 64; ++r) {
#pragma unroll
      for (int k = 0; k < 16; ++k) v[k] = p[(k * 7 + r) & 63];
#pragma unroll
      for (int k = 0; k < 16; ++k) asm volatile("" : "+v"(v[k]));
    }
    x = v[0][0] + v[15][1];
  } else if (OP == 9) {      // dependent v_cndmask pair (a double select) -> fma
#pragma unroll
    for (int k = 0; k < N / 4; ++k) {
      x = (lane & 1) ? x : y;
      x = fma(x, a, b);
    }
  }
  asm volatile("" : "+v"(x));
#pragma unroll
  for (int q = 0; q < 8; ++q) asm volatile("" : "+v"(acc[q]));
  const long long c1 = clock64();
  const long long w1 = wall_clock64();
  asm volatile("" : "+v"(x));
  double r = x;
#pragma unroll
  for (int q = 0; q < 8; ++q) r += acc[q];
  out[blockIdx.x * 256 + threadIdx.x] = r;
  if (threadIdx.x == 0) {
    ticks[0] = c1 - c0;
    ticks[1] = w1 - w0;
  }
}

int main() {
  double* d;
  long long* t;
  CK(hipMalloc(&d, 256 * 8));
  CK(hipMalloc(&t, 16));
  long long h[2];
  const char* names[] = {"dependent v_fma_f64", "8 independent v_fma_f64 chains", "v_rsq_f64 + correction + add (8 dependent ops)",
                         "v_readlane x2 -> v_add_f64", "LDS write -> read(other lane) -> add", "DPP quad bcast x2 -> fma",
                         "v_permlane32_swap x2 -> fma", "independent ds_read_b128 + 2 adds", "dependent v_mul_f64", "select(2 cndmask) -> fma",
                         "2 interleaved dependent v_fma_f64", "4 interleaved dependent v_fma_f64", "independent ds_read_b128 (const offsets)"};
  const int per[] = {N, N, N / 8, N / 4, N / 8, N / 4, N / 4, N / 4, N, N / 4, N, N, N / 4};
#define RUN(OP, NT)                                                                                   \
  do {                                                                                                \
    for (int pass = 0; pass < 2; ++pass) {                                                            \
      hipLaunchKernelGGL(micro<OP>, dim3(1), dim3(NT), 0, 0, d, t, 1.25);                             \
      CK(hipDeviceSynchronize());                                                                     \
    }                                                                                                 \
    CK(hipMemcpy(h, t, 16, hipMemcpyDeviceToHost));                                                   \
    std::printf("%-48s %3d thr: %7.1f shader clocks, %6.1f ns per iteration (%d iterations; %.2f GHz)\n", names[OP], NT, \
                (double)h[0] / per[OP], 10.0 * h[1] / per[OP], per[OP], h[1] ? h[0] / (10.0 * h[1]) : 0.0);   \
  } while (0)
  RUN(0, 64); RUN(0, 256);
  RUN(1, 64); RUN(1, 256);
  RUN(2, 64);
  RUN(3, 64);
  RUN(4, 64); RUN(4, 256);
  RUN(5, 64);
  RUN(6, 64);
  RUN(7, 64); RUN(7, 256);
  RUN(8, 64);
  RUN(9, 64);
  RUN(10, 64);
  RUN(11, 64);
  RUN(12, 64); RUN(12, 256);
  return 0;
}
