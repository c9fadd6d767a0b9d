// Device-side helpers shared by the gfx950 kernel files (kernels.hip, mg.hip): launch geometry, the XCD-aware block order and the
// in-launch grid reductions.  HIP-only header (included from .hip translation units).
#pragma once
#include <hip/hip_runtime.h>

#include "kernels.hpp"

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

// XCD-aware block order for the gathering kernels.  Workgroups are dealt round-robin over the 8 XCDs (blocks b and b + 8
// share one, MI355X_MICROARCH.md) and every XCD has its own 4 MiB L2: with the natural order, neighbouring row blocks --
// which gather the same entries of x -- land on eight different L2s and every one of them fetches those entries again
// (measured at fem2d L=9: restriction 2.6x, Hessian assembly 2.4x, apply_D 1.3x the algorithmic bytes at the fabric,
// profiles/r2_probe_L9_pmc_traffic.json).  Remapped, XCD k works through the k-th contiguous eighth of the blocks.
// Speed only: any placement gives the same result.
__device__ inline unsigned xcd_block(unsigned b, unsigned nb) {
  if (nb < 16u) return b;
  const unsigned per = nb >> 3, main = per << 3;      // blocks beyond a multiple of 8 keep their place
  return b < main ? (b & 7u) * per + (b >> 3) : b;
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

// N block sums at once: one pair of barriers instead of N (per value the additions and their order are those of block_sum);
// results valid in thread 0
template <int N>
__device__ inline void block_sum_n(const double (&v)[N], double (&out)[N]) {
  __shared__ double red[N][kBlock / 64];
  const int wave = threadIdx.x >> 6;
#pragma unroll
  for (int o = 0; o < N; ++o) {
    const double w = wave_sum(v[o]);
    if ((threadIdx.x & 63) == 0) red[o][wave] = w;
  }
  __syncthreads();
#pragma unroll
  for (int o = 0; o < N; ++o) {
    double r = 0.0;
    if (threadIdx.x == 0)
      for (int i = 0; i < kBlock / 64; ++i) r += red[o][i];
    out[o] = r;
  }
  __syncthreads();
}

// Grid-wide sum of per-block partial results WITHOUT a second launch: thread 0 of every block publishes its `NOUT` values
// (write-through sc1 stores, drained), then takes a ticket; the block whose ticket is the last one reads all partials
// back (sc1 loads, bypassing its L1) and sums them in a FIXED order -- thread-strided, then the block tree -- so the
// result does not depend on which block finishes last.  This is the "agent-scope atomic add by one lane of each storing
// workgroup, last adder consumes" hand-off of MI355X_MICROARCH.md (Valid forms; no L2 write-back, no L1 invalidate).
// One counter serialises its arrivals (~12 ns each: 11 us for the 896 blocks of the fused objective kernel at fem2d L=7,
// measured), so the ticket is two-level: 8 shard counters (block id mod 8: one XCD each under round-robin placement, for
// speed only) on cache lines of their own, whose last arrivers meet on a top counter.
// scratch layout: kTicketDoubles doubles of ticket words (zero between launches), then the partials.
// out_dev / out_host (either may be null): device result for a following collective, pinned host memory for the host.
constexpr int kTicketDoubles = kReductionHeader;      // 9 counters, one 128-byte line each (atomics on one line serialise)
constexpr int kTicketStride = 32;                     // unsigned words per line
static_assert(kReductionHeader * sizeof(double) >= 9 * kTicketStride * sizeof(unsigned), "ticket words");
template <int NOUT>
__device__ inline void grid_finish(const double (&r)[NOUT] /* valid in thread 0 */, double* scratch, double* out_dev,
                                   double* out_host, double* lds, HostSignal sig = HostSignal(), unsigned slot = 0xffffffffu) {
  if (slot == 0xffffffffu) slot = blockIdx.x;      // position of this block's partial in the fixed summation order
  __shared__ int is_last;
  unsigned* ticket = reinterpret_cast<unsigned*>(scratch);      // counter k lives at ticket[kTicketStride * k]: a 128-byte line each
  double* partials = scratch + kTicketDoubles;
  if (threadIdx.x == 0) {
#pragma unroll
    for (int o = 0; o < NOUT; ++o)
      __hip_atomic_store(&partials[(size_t)slot * NOUT + o], r[o], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned shard = blockIdx.x & 7u, nshards = gridDim.x < 8u ? gridDim.x : 8u;
    const unsigned in_shard = (gridDim.x - shard + 7u) / 8u;      // blocks with this shard id
    bool last = false;
    if (__hip_atomic_fetch_add(&ticket[kTicketStride * shard], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == in_shard - 1) {
      __hip_atomic_store(&ticket[kTicketStride * shard], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // re-arm
      last = __hip_atomic_fetch_add(&ticket[kTicketStride * 8], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nshards - 1;
      if (last) __hip_atomic_store(&ticket[kTicketStride * 8], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    is_last = last;
  }
  __syncthreads();
  if (!is_last) return;      // workgroup-uniform
  // all NOUT values of a block are read in one go (the loads bypass this XCD's L2: one memory round trip per batch; a loop
  // over the outputs paid NOUT of them one after the other, 6 in the fused objective kernel); per output the additions keep
  // their order (thread-strided, then the block tree)
  double acc[NOUT];
#pragma unroll
  for (int o = 0; o < NOUT; ++o) acc[o] = 0.0;
  for (int i0 = threadIdx.x; i0 < (int)gridDim.x; i0 += 4 * kBlock) {      // four blocks' values per batch (index clamped, sum selected)
    double v[4][NOUT];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = min(i0 + u * kBlock, (int)gridDim.x - 1);
#pragma unroll
      for (int o = 0; o < NOUT; ++o) v[u][o] = __hip_atomic_load(&partials[(size_t)i * NOUT + o], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int o = 0; o < NOUT; ++o) acc[o] = (i0 + u * kBlock < (int)gridDim.x) ? acc[o] + v[u][o] : acc[o];
  }
  double tot[NOUT];
  block_sum_n<NOUT>(acc, tot);
  (void)lds;
  if (threadIdx.x == 0) {
#pragma unroll
    for (int o = 0; o < NOUT; ++o) {
      if (out_dev) out_dev[o] = tot[o];
      if (out_host) __hip_atomic_store(&out_host[o], tot[o], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
  if (sig.seq_host && threadIdx.x == 0) {      // results first (drained), then the sequence number the host polls
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long q = __hip_atomic_load(sig.seq_dev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1;
    __hip_atomic_store(sig.seq_dev, q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(sig.seq_host, q, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}
// The same hand-off with the finishing arithmetic left to the caller: `fn(tot)` runs in thread 0 of the last block with the
// NOUT sums (fixed summation order, as grid_finish), e.g. to turn two dots into a step length that the next launch reads
// from device memory -- no host round trip inside a Krylov iteration.
template <int NOUT, class Fn>
__device__ inline void grid_finish_fn(const double (&r)[NOUT] /* valid in thread 0 */, double* scratch, double* lds, Fn&& fn) {
  __shared__ int is_last_fn;
  unsigned* ticket = reinterpret_cast<unsigned*>(scratch);
  double* partials = scratch + kTicketDoubles;
  if (threadIdx.x == 0) {
#pragma unroll
    for (int o = 0; o < NOUT; ++o)
      __hip_atomic_store(&partials[(size_t)blockIdx.x * NOUT + o], r[o], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned shard = blockIdx.x & 7u, nshards = gridDim.x < 8u ? gridDim.x : 8u;
    const unsigned in_shard = (gridDim.x - shard + 7u) / 8u;
    bool last = false;
    if (__hip_atomic_fetch_add(&ticket[kTicketStride * shard], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == in_shard - 1) {
      __hip_atomic_store(&ticket[kTicketStride * shard], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      last = __hip_atomic_fetch_add(&ticket[kTicketStride * 8], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nshards - 1;
      if (last) __hip_atomic_store(&ticket[kTicketStride * 8], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    is_last_fn = last;
  }
  __syncthreads();
  if (!is_last_fn) return;      // workgroup-uniform
  double tot[NOUT], acc[NOUT];
#pragma unroll
  for (int o = 0; o < NOUT; ++o) acc[o] = 0.0;
  for (int i0 = threadIdx.x; i0 < (int)gridDim.x; i0 += 4 * kBlock) {      // one round trip per batch of 4 NOUT loads (see grid_finish)
    double v[4][NOUT];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = min(i0 + u * kBlock, (int)gridDim.x - 1);
#pragma unroll
      for (int o = 0; o < NOUT; ++o) v[u][o] = __hip_atomic_load(&partials[(size_t)i * NOUT + o], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int o = 0; o < NOUT; ++o) acc[o] = (i0 + u * kBlock < (int)gridDim.x) ? acc[o] + v[u][o] : acc[o];
  }
  block_sum_n<NOUT>(acc, tot);
  (void)lds;
  if (threadIdx.x == 0) fn(tot);
}

}  // namespace
}  // namespace mgb
