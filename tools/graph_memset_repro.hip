// Reproducer for the hipGraph memset-node observation of round 1 (gpurun_out/sh5.log: "pivot flag 808464432" =
// 0x30303030, a byte fill that only a memset can produce -- the kernels only ever atomicOr(flag, 1)).
// A 4-byte hipMemsetAsync(ptr, 0, 4) is captured into a graph together with a kernel that ORs 1 into the word;
// the graph is replayed while the host heap is churned with '0'-filled strings between replays (what a Python
// host does all the time).  Prints the first flag value that is neither 0 nor 1.
//   hipcc --offload-arch=gfx950 -O2 tools/graph_memset_repro.hip -o gpurun_out/graph_memset_repro && gpurun_out/graph_memset_repro
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

__global__ void maybe_or(int* flag, int doit) {
  if (doit && threadIdx.x == 0) atomicOr(flag, 1);
}

#define CK(x)                                                                  \
  do {                                                                         \
    hipError_t e = (x);                                                        \
    if (e != hipSuccess) {                                                     \
      std::printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); \
      return 2;                                                                \
    }                                                                          \
  } while (0)

int main() {
  hipStream_t st;
  CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  int* d = nullptr;
  CK(hipMalloc((void**)&d, sizeof(int)));
  int* h = nullptr;
  CK(hipHostMalloc((void**)&h, sizeof(int), hipHostMallocDefault));
  hipGraph_t g;
  hipGraphExec_t ge;
  CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
  CK(hipMemsetAsync(d, 0, sizeof(int), st));
  hipLaunchKernelGGL(maybe_or, dim3(1), dim3(64), 0, st, d, 0);
  CK(hipStreamEndCapture(st, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  CK(hipGraphDestroy(g));
  int bad = 0;
  for (int it = 0; it < 20000 && !bad; ++it) {
    // heap churn between replays: short-lived buffers full of ASCII '0'
    std::vector<std::string> junk;
    for (int k = 0; k < 8; ++k) junk.emplace_back((size_t)(16 + 8 * ((it + k) % 64)), '0');
    CK(hipGraphLaunch(ge, st));
    CK(hipMemcpyAsync(h, d, sizeof(int), hipMemcpyDeviceToHost, st));
    CK(hipStreamSynchronize(st));
    if (*h != 0 && *h != 1) {
      std::printf("replay %d: flag = %d (0x%08x)\n", it, *h, (unsigned)*h);
      bad = 1;
    }
    // dirty the word from outside the graph, as a previous non-SPD factorisation would
    CK(hipMemsetAsync(d, 0x7f, sizeof(int), st));
  }
  std::printf(bad ? "graph memset node wrote a wrong fill pattern\n" : "20000 replays: the graph memset node always wrote 0\n");
  return 0;
}
