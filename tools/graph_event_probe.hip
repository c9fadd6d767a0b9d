// Can HIP events recorded INSIDE a captured graph time single kernel nodes on replay?  (bench.py's roofline object wants
// per-kernel durations in the launch mode the solve really uses: hipGraph replay of the factorisation chain.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void spin(long long cycles, int* sink) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < cycles) {}
  if (sink && threadIdx.x == 12345) *sink = 1;
}
#define CK(x) do { hipError_t err_ = (x); if (err_ != hipSuccess) { std::printf("HIP error %s at line %d\n", hipGetErrorString(err_), __LINE__); return 2; } } while (0)
int main() {
  hipStream_t st;
  CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  const int n = 6;
  std::vector<hipEvent_t> ev(n + 1);
  for (auto& e : ev) CK(hipEventCreate(&e));
  hipGraph_t g; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
  for (int k = 0; k < n; ++k) {
    CK(hipEventRecord(ev[k], st));
    hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, st, (long long)(500 * (k + 1)), nullptr);   // 100 MHz clock: 5, 10, ... us
  }
  CK(hipEventRecord(ev[n], st));
  CK(hipStreamEndCapture(st, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipGraphLaunch(ge, st));
    CK(hipStreamSynchronize(st));
    std::printf("replay %d:", rep);
    for (int k = 0; k < n; ++k) {
      float ms = -1;
      hipError_t e = hipEventElapsedTime(&ms, ev[k], ev[k + 1]);
      std::printf(" %s%.2f", e == hipSuccess ? "" : "ERR", ms * 1e3);
    }
    std::printf(" us (expected 5 10 15 20 25 30 + boundary)\n");
  }
  return 0;
}
