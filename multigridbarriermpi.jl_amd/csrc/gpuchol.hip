// gfx950 kernels + host schedule of the device multifrontal Cholesky (see gpuchol.hpp).
//
// Roofline: at the benchmark sizes this solver is bound by the LENGTH of its dependent launch chain (one
// launch per 32 pivot columns along the tallest root-to-leaf path of the elimination tree), not by HBM or
// the fp64 pipes: 0.35 GFLOP and 106 MB of fronts per fine-level factorisation at fem2d L=7, all of it
// L2 / Infinity-Cache resident.  The design therefore minimises launches and dependent global-memory
// round trips per launch (descriptor -> operands -> results), and spends redundant flops freely (every
// tile re-derives its two 64x32 panel blocks instead of waiting for a separate TRSM launch).  The rank-32
// trailing updates run on the fp64 matrix cores (v_mfma_f64_16x16x4_f64): same peak as the vector unit on
// MI355X, but one instruction per 1 024 multiply-adds and a quarter of the LDS operand traffic, which is
// what counts on larger meshes (fem2d L >= 8, fem3d) where the launches are throughput-bound.
#include "gpuchol.hpp"

#include "amg.hpp"

#include <algorithm>
#include <array>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <stdexcept>
#include <string>

namespace mgb {

namespace {

constexpr int PB = 32;       // panel width
constexpr int LP = PB + 1;   // padded LDS row length
constexpr int TS = 64;       // trailing-update tile
constexpr int TB = 256;      // threads per workgroup (factorisation kernels)
constexpr int RT = 1024;     // threads per workgroup (backward_rect)

void ck(hipError_t e, const char* what) {
  if (e != hipSuccess) throw HipError(std::string("HIP error in gpuchol ") + what + ": " + hipGetErrorString(e));
}

// optional phase stamps (100 MHz wall clock) of workgroup 0 of every factorisation launch: MGB_CHOL_PROF=1
// slots 0..7: phase stamps of workgroup 0.  With -DMGB_PROF_ALL_WGS (a debugging build: the atomics cost registers in every
// kernel) slot 8 / 9 also record the earliest STAMP(0) / latest STAMP(7) over ALL workgroups of the launch and which
// workgroup finished last -- how the off-diagonal tiles were found to bound a panel step (profiles/r2_chol_phase_stamps_L7.txt).
constexpr int kProfSlots = 10;
// -DMGB_PROF_LAST_WG (debugging build): the stamps follow the LAST workgroup of a launch (a trailing-matrix tile) instead
// of workgroup 0 (the pivot workgroup of the first front)
#ifdef MGB_PROF_LAST_WG
#define MGB_PROF_WG (gridDim.x - 1)
#else
#define MGB_PROF_WG 0
#endif
// -DMGB_PROF_PER_WG (with -DMGB_PROF_ALL_WGS): start and end of EVERY workgroup of every stamped launch, printed as a
// per-launch summary (dispatch stagger, duration spread, the slowest workgroups)
#ifdef MGB_PROF_PER_WG
constexpr int kProfMaxWg = 2048;
__device__ long long* g_wgprof = nullptr;
__device__ long long* g_prof_base = nullptr;
#define STAMP_WG(k, now_)                                                                                               \
  if (((k) == 0 || (k) == 7) && g_wgprof && blockIdx.x < kProfMaxWg)                                                     \
    g_wgprof[(((prof - g_prof_base) / kProfSlots) * kProfMaxWg + blockIdx.x) * 2 + ((k) == 7)] = now_;
#else
#define STAMP_WG(k, now_)
#endif
#ifdef MGB_PROF_ALL_WGS
#define STAMP(k)                                                                                   \
  do {                                                                                             \
    if (prof && threadIdx.x == 0) {                                                                \
      const long long now_ = wall_clock64();                                                       \
      STAMP_WG(k, now_)                                                                            \
      if (blockIdx.x == MGB_PROF_WG) prof[k] = now_;                                               \
      if ((k) == 0) atomicMin((unsigned long long*)&prof[8], (unsigned long long)now_);            \
      if ((k) == 7) atomicMax((unsigned long long*)&prof[9], (unsigned long long)now_ * 65536ull + (blockIdx.x & 65535u)); \
    }                                                                                              \
  } while (0)
#else
#define STAMP(k)                                                                   \
  do {                                                                             \
    if (prof && blockIdx.x == MGB_PROF_WG && threadIdx.x == 0) prof[k] = wall_clock64(); \
  } while (0)
#endif

// broadcast lane `lane` (compile-time constant) of a double through SGPRs
__device__ inline double readlane_f64(double v, int lane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}

// Cholesky factor L of the kw x kw lower-triangular block D (LDS, D[i*LP+j], j <= i; padded with the identity
// up to 32x32) by wave 0, in registers, TWO pivot columns per round.  Lane (i, h) owns the entries of row i in
// the column pairs (4m + 2h, 4m + 2h + 1).  Per round the three pivot-block entries are broadcast with
// constant-lane v_readlane, every lane forms 1/L[k][k], L[k+1][k], 1/L[k+1][k+1], the owning half computes its
// two multipliers per row, and these go through a 64-entry LDS line (one write, then conflict-free broadcast
// reads; LDS executes a wave's accesses in order, so no barrier is needed): 16 LDS round trips per block
// instead of a v_readlane/s_nop pair per ENTRY.  Entries right of the diagonal of a row are don't-care (a finished row
// keeps "updating" them with whatever its multiplier slot holds), so the update is branch-free.  The diagonal slot keeps
// 1/L[k][k] (L[k][k] itself is never needed again): every consumer (panel TRSM, backward sweep) is a
// substitution that multiplies by it.  Stores the block row-major to lp[32 i + j] and column-major behind it
// (zero above the diagonal).
// Called by every thread of the workgroup (contains a barrier); D must be visible (barrier) before the call.
// Lo: 32 x LP LDS scratch, distinct from D.
// LDS pointers are passed in their own address space: through a generic `double*` every access re-derives the LDS
// address with a null check (v_cmp + v_cndmask per read), which costs issue slots on the single-wave critical path
using lds_f64 = __attribute__((address_space(3))) double*;
using lds_cf64 = const __attribute__((address_space(3))) double*;

// 1 / sqrt(x): v_rsq_f64 and one Newton-type correction, the arithmetic of the device library's rsqrt() without its
// fix-up of x = 0 / inf (three more instructions on the single-wave critical path; such pivots are flagged by the caller)
__device__ __forceinline__ double rsq_refined(double x) {
  const double y = __builtin_amdgcn_rsq(x);
  const double e = fma(y * -x, y, 1.0);
  return fma(y * e, fma(e, 0.375, 0.5), y);
}

// the 16 rounds of the in-register factorisation: straight-line code, or (BRK) with an exit test per round.
// ONE wave issues everything here, one instruction per ~4.5 cycles, and the rounds were measured at the product of the two
// (675 cycles = ~135 instructions per round), so the code is written for instruction count:
//   * no pivot test inside the round: a pivot that is not positive and finite turns its reciprocal root (and everything
//     after it) into NaN / inf, which `chk` collects with four instructions per round (x * 0 is NaN for both);
//   * rows above the pivot keep whatever the update leaves in their don't-care entries right of the diagonal (no
//     multiplier is forced to zero): nothing reads those, the store masks them;
//   * the half of the wave that does not own the pivot columns is switched off by the execution mask instead of
//     selecting per value.
template <bool BRK>
__device__ inline void factor_rounds(double (&a)[PB / 2], lds_f64 col, int i, int h, double& chk, int kw) {
  // No control flow inside a round (the compiler keeps a[] as one register tuple and copied all of it around every
  // masked region): the multipliers are computed by every lane and SELECTED by the owning half, the other half writes
  // its (meaningless) pair to a dummy line behind the broadcast line, and the update of the h = 1 lanes' pivot-slot
  // columns in the even rounds is switched off in the h = 0 lanes by a zero multiplier.
  // LDS: col[0..63] multipliers (row i at 2 i), col[64..127] dummy line, col[128..159] the reciprocal pivots.
  const bool h0 = h == 0;
  const double hm = h0 ? 0.0 : 1.0;
  lds_f64 wr[2] = {col + (h0 ? 0 : 64) + 2 * i, col + (h0 ? 64 : 0) + 2 * i};      // where this lane writes in even / odd rounds
#pragma unroll
  for (int kk = 0; kk < PB / 2; ++kk) {
    const int k = 2 * kk, hk = kk & 1, mk = kk >> 1;
    if (BRK && k >= kw) break;
    const double akk = readlane_f64(a[2 * mk], k + 32 * hk);
    const double ak1k = readlane_f64(a[2 * mk], k + 1 + 32 * hk);
    const double ak1k1 = readlane_f64(a[2 * mk + 1], k + 1 + 32 * hk);
    const double rd0 = rsq_refined(akk);        // 1 / L[k][k]
    const double l10 = ak1k * rd0;              // L[k+1][k]
    const double d1 = fma(-l10, l10, ak1k1);
    const double rd1 = rsq_refined(d1);         // 1 / L[k+1][k+1]
    chk = fma(akk + d1, 0.0, chk);
    chk = fma(rd0 + rd1, 0.0, chk);
    const double l0 = a[2 * mk] * rd0;
    const double l1 = fma(-l0, l10, a[2 * mk + 1]) * rd1;
    wr[hk][0] = l0;
    wr[hk][1] = l1;
    col[128 + k] = rd0;       // every lane, the same values: the diagonal slots are patched from here after the rounds
    col[128 + k + 1] = rd1;
    const bool own = hk ? !h0 : h0;
    a[2 * mk] = own ? l0 : a[2 * mk];
    a[2 * mk + 1] = own ? l1 : a[2 * mk + 1];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    const double li0 = col[2 * i], li1 = col[2 * i + 1];
    if (hk == 0) {      // the pair (k+2, k+3) lives in the h = 1 lanes at the same register slots
      const double m0 = li0 * hm, m1 = li1 * hm;
      a[2 * mk] = fma(-m1, col[2 * (k + 2) + 1], fma(-m0, col[2 * (k + 2)], a[2 * mk]));
      a[2 * mk + 1] = fma(-m1, col[2 * (k + 3) + 1], fma(-m0, col[2 * (k + 3)], a[2 * mk + 1]));
    }
#pragma unroll
    for (int m = mk + 1; m < PB / 4; ++m) {
      const int c = 4 * m + 2 * h;
      a[2 * m] = fma(-li1, col[2 * c + 1], fma(-li0, col[2 * c], a[2 * m]));
      a[2 * m + 1] = fma(-li1, col[2 * c + 3], fma(-li0, col[2 * c + 2], a[2 * m + 1]));
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// EXIT (front_leaf, whose second panel has 2..11 columns; front_single launches whose fronts have <= 8 pivots): leave
// the rounds at the identity padding, which needs none (bitwise the same result).  The exit test per round costs a
// full-width block 4 us (measured on front_start / front_step), and 4/8/16-round variants behind one branch made
// front_single slower whenever two of its workgroups shared a CU, so everything else runs the plain 16 rounds.
// PACKED (front_leaf): Lo is the packed lower triangle (row r at r (r + 1) / 2, 528 doubles): with the packed front
// this is what lets four leaf workgroups share a CU.
__device__ inline constexpr int lo_packed(int r, int c) { return r * (r + 1) / 2 + c; }

template <bool EXIT = false, bool PACKED = false>
__device__ __forceinline__ void factor_diag_block(const double* D, int kw, double* Lo, double* lp, int* fail, long long* prof) {
  const int tid = threadIdx.x;
  if (tid < 64) {
    const int i = tid & 31, h = tid >> 5;
    double a[PB / 2];
#pragma unroll
    for (int q = 0; q < PB / 2; ++q) {
      const int j = 4 * (q >> 1) + 2 * h + (q & 1);
      double v = (i == j) ? 1.0 : 0.0;
      if (i < kw && j < kw && j <= i) v = D[i * LP + j];
      a[q] = v;
    }
    double chk = 0.0;
    lds_f64 col = (lds_f64)Lo;      // the broadcast line aliases the output block, which is only written after the loop
    if (EXIT) factor_rounds<true>(a, col, i, h, chk, kw);
    else factor_rounds<false>(a, col, i, h, chk, PB);
    STAMP(6);
    if (chk != chk && tid == 0) atomicOr(fail, 1);
    // the reciprocal pivot of this lane's row (rows of the identity padding the rounds never reached keep their 1)
    const double rdv = (!EXIT || i < ((kw + 1) & ~1)) ? col[128 + i] : 1.0;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int q = 0; q < PB / 2; ++q) {
      const int j = 4 * (q >> 1) + 2 * h + (q & 1);
      if (PACKED) {
        if (j <= i) Lo[lo_packed(i, j)] = a[q];
      } else {
        Lo[i * LP + j] = (j <= i) ? a[q] : 0.0;
      }
    }
    // the diagonal slot holds 1 / L[i][i]: patched behind the row stores (LDS keeps a wave's accesses in order)
    Lo[PACKED ? lo_packed(i, i) : i * LP + i] = rdv;
  }
  __syncthreads();
  if (lp == nullptr) return;      // workgroup-uniform: the caller keeps the factor in Lo only
  for (int idx = tid; idx < PB * PB; idx += blockDim.x) {
    const int r = idx / PB, c = idx % PB;
    if (PACKED) {
      lp[idx] = (c <= r) ? Lo[lo_packed(r, c)] : 0.0;
      lp[PB * PB + idx] = (r <= c) ? Lo[lo_packed(c, r)] : 0.0;
    } else {
      lp[idx] = Lo[r * LP + c];               // row-major: lp[32 m + c] = L[m][c]
      lp[PB * PB + idx] = Lo[c * LP + r];     // column-major copy: L[m][j] at 32 j + m
    }
  }
}

// The wave-0 part of factor_diag_block without its barriers (front_step2: the other three waves work on their panel
// blocks meanwhile): D (LDS, row stride LP) -> Lo[i * LP + j], reciprocal diagonal, zeros above it.  Threads 0..63 only.
__device__ __forceinline__ void factor_block_wave0(const double* D, int kw, double* Lo, int* fail) {
  const int tid = threadIdx.x, i = tid & 31, h = tid >> 5;
  double a[PB / 2];
#pragma unroll
  for (int q = 0; q < PB / 2; ++q) {
    const int j = 4 * (q >> 1) + 2 * h + (q & 1);
    double v = (i == j) ? 1.0 : 0.0;
    if (i < kw && j < kw && j <= i) v = D[i * LP + j];
    a[q] = v;
  }
  double chk = 0.0;
  factor_rounds<false>(a, (lds_f64)Lo, i, h, chk, PB);
  if (chk != chk && tid == 0) atomicOr(fail, 1);
  const double rdv = ((lds_f64)Lo)[128 + i];
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int q = 0; q < PB / 2; ++q) {
    const int j = 4 * (q >> 1) + 2 * h + (q & 1);
    Lo[i * LP + j] = (j <= i) ? a[q] : 0.0;
  }
  Lo[i * LP + i] = rdv;
}

// Start of a height: workgroup = 32 columns x 256 rows of one front, in gather form.  (1) every lower entry (and the
// right-hand-side row nf) of the columns is WRITTEN as child0 + child1 contribution (or 0), found through the
// per-child inverse index maps (parent row -> child boundary row, -1 if absent): no read-modify-write chains,
// the loads of all 32 columns are issued (branch-free) before their stores; (2) the assembled matrix entries and the right-hand side
// are added; (3) the workgroup of columns 0..31 factors the first pivot block.  Every parent entry belongs to
// exactly one workgroup and the order of the additions is fixed.
__global__ __launch_bounds__(TB, 3) void front_start_kernel(const GNode* __restrict__ nodes, const StartJob* __restrict__ jobs,
                                                          const int* __restrict__ pinv, const int* __restrict__ asm_src,
                                                          const int* __restrict__ asm_pos, const double* __restrict__ vals,
                                                          const int* __restrict__ perm, const double* __restrict__ b,
                                                          const double* __restrict__ fronts_ro, double* fronts, double* linv,
                                                          int* fail, long long* prof, int dedicated_pivot) {
  __shared__ double sh[2 * PB * LP];
  __shared__ int cb[2][PB];
  STAMP(0);
  const StartJob job = jobs[blockIdx.x];      // self-contained: node, children, assembly range
  const int nf = job.nf, ld = nf + 1, ns = job.ns;
  double* F = fronts + job.off;
  const int c0 = job.chunk * PB, c1 = min(nf, c0 + PB);
  const int tid = threadIdx.x;
  // child blocks (read-only here, through fronts_ro): entry (boundary row a, boundary column q) of child s at
  // fronts_ro[boff[s] + cld[s] * q + a]; an absent child gets a valid dummy offset that is never selected
  const long long boff[2] = {job.boff[0], job.boff[1]};
  const int cld[2] = {job.cld[0], job.cld[1]};
  const bool has[2] = {job.cld[0] > 0, job.cld[1] > 0};
  const int* __restrict__ inv0 = pinv + (has[0] ? job.iofs : 0);
  const int* __restrict__ inv1 = pinv + (has[1] ? job.iofs + ld : 0);
  if (job.rb < 0) {
    // Dedicated pivot job (one per front, first in the launch): the first pivot block is gathered straight into LDS --
    // child0 + child1, then the matrix entries, the order of the assembling workgroups -- and factored at once, instead
    // of waiting for the assembly of 256 x 32 entries, its read-modify-write of the matrix entries in global memory and
    // the reload of the block (two round trips on what the first panel launch waits for).  Bitwise the same block.
    const int kw = min(PB, ns);
    double* D = sh;
    int apos[2], asrc[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int k = job.a0 + u * TB + tid;
      apos[u] = (k < job.a1) ? asm_pos[k] : -1;
      asrc[u] = (k < job.a1) ? asm_src[k] : 0;
    }
    if (tid < 2 * PB) {
      const int s = tid / PB, c = tid % PB;
      cb[s][c] = (has[s] && c < kw) ? (s ? inv1 : inv0)[c] : -1;
    }
    double aval[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) aval[u] = vals[asrc[u]];
    __syncthreads();
    STAMP(1);
    const double* __restrict__ B0 = fronts_ro + boff[0];
    const double* __restrict__ B1 = fronts_ro + boff[1];
    {
      const int i = tid % PB, jg = tid / PB;      // 32 rows x 8 groups of 4 columns
      double v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int j = jg * 4 + u;
        const int a0 = cb[0][i], q0 = cb[0][j], a1 = cb[1][i], q1 = cb[1][j];
        const bool in = (j <= i && i < kw), ok0 = in && a0 >= 0 && q0 >= 0, ok1 = in && a1 >= 0 && q1 >= 0;
        const double x0 = B0[ok0 ? (long long)cld[0] * q0 + a0 : 0];
        const double x1 = B1[ok1 ? (long long)cld[1] * q1 + a1 : 0];
        v[u] = (ok0 ? x0 : 0.0) + (ok1 ? x1 : 0.0);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) D[i * LP + jg * 4 + u] = v[u];
    }
    __syncthreads();
    STAMP(2);
    auto add_entry = [&](int pos, double v) {
      if (pos < 0) return;
      const int col = pos / ld, row = pos - col * ld;
      if (row < PB && col < PB) D[row * LP + col] += v;
    };
#pragma unroll
    for (int u = 0; u < 2; ++u) add_entry(apos[u], aval[u]);
    for (int k = job.a0 + 2 * TB + tid; k < job.a1; k += TB) add_entry(asm_pos[k], vals[asm_src[k]]);
    __syncthreads();
    STAMP(5);
    factor_diag_block(D, kw, sh + PB * LP, linv + job.loff, fail, prof);
    STAMP(7);
    return;
  }
  const int i = c0 + job.rb * TB + tid;        // one row per thread: the workgroup owns rows [c0 + 256 rb, +256)
  // everything that only needs the descriptor is requested together: the column maps, this thread's row maps, the first
  // assembly indices and the right-hand-side permutation (the chain is descriptor -> indices -> data)
  if (tid < 2 * PB) {
    const int s = tid / PB, c = c0 + tid % PB;
    cb[s][tid % PB] = (has[s] && c < c1) ? (s ? inv1 : inv0)[c] : -1;
  }
  const int ra0 = (i <= nf && has[0]) ? inv0[i] : -1, ra1 = (i <= nf && has[1]) ? inv1[i] : -1;
  const int ka = job.a0 + tid;
  const int apos0 = (ka < job.a1) ? asm_pos[ka] : -1, asrc0 = (ka < job.a1) ? asm_src[ka] : 0;
  const bool rhs_wg = (c0 + job.rb * TB <= nf && nf < c0 + (job.rb + 1) * TB);      // this workgroup owns the right-hand-side row
  const int crhs = c0 + tid;
  const int prm = (rhs_wg && crhs < min(c1, ns)) ? perm[job.first + crhs] : -1;
  const double aval0 = vals[asrc0];
  const double bval = (prm >= 0) ? b[prm] : 0.0;
  __syncthreads();
  STAMP(1);
  const double* __restrict__ B0 = fronts_ro + boff[0];
  const double* __restrict__ B1 = fronts_ro + boff[1];
  if (i <= nf && !has[0] && !has[1]) {      // leaf front (workgroup-uniform): nothing to gather, just clear
    const int cend = min(c1, i + 1);
#pragma unroll
    for (int q = 0; q < PB; ++q)
      if (c0 + q < cend) F[(long long)ld * (c0 + q) + i] = 0.0;
  } else if (i <= nf) {
    const int cend = min(c1, i + 1);       // lower triangle: columns c <= i (row nf: every column)
    double v[PB];
    // branch-free: every load is issued (clamped to a valid address) before any is consumed
#pragma unroll
    for (int q = 0; q < PB; ++q) {
      const int q0 = cb[0][q], q1 = cb[1][q];
      const bool ok0 = (c0 + q < cend) && ra0 >= 0 && q0 >= 0, ok1 = (c0 + q < cend) && ra1 >= 0 && q1 >= 0;
      const double x0 = B0[ok0 ? (long long)cld[0] * q0 + ra0 : 0];
      const double x1 = B1[ok1 ? (long long)cld[1] * q1 + ra1 : 0];
      v[q] = (ok0 ? x0 : 0.0) + (ok1 ? x1 : 0.0);
    }
#pragma unroll
    for (int q = 0; q < PB; ++q)
      if (c0 + q < cend) F[(long long)ld * (c0 + q) + i] = v[q];
  }
  __syncthreads();
  STAMP(2);
  if (apos0 >= 0) F[apos0] += aval0;      // first batch from registers, the rest (large fronts) the long way
  for (int k = job.a0 + TB + tid; k < job.a1; k += TB) F[asm_pos[k]] += vals[asm_src[k]];
  if (prm >= 0) F[(long long)ld * crhs + nf] += bval;      // min(c1, ns) - c0 <= 32 <= TB columns: one per thread
  if (job.chunk == 0 && job.rb == 0 && ns > 0 && !dedicated_pivot) {      // (the schedule without pivot jobs)
    __syncthreads();
    STAMP(3);
    const int kw = min(PB, ns);
    double* D = sh;
    for (int idx = tid; idx < PB * PB; idx += TB) {
      const int i = idx % PB, j = idx / PB;
      D[i * LP + j] = (i < kw && j <= i) ? F[(long long)ld * j + i] : 0.0;
    }
    __syncthreads();
    STAMP(5);
    factor_diag_block(D, kw, sh + PB * LP, linv + job.loff, fail, prof);
  }
  STAMP(7);
}

// Panel p of every front of a height that has one.  Workgroup = 64x64 tile (ti, tj) of the trailing matrix
// (rows / columns counted from k1 = end of the panel; rows run to nf INCLUSIVE: the right-hand-side row).
//   L_I = A[I, panel] * L11^-T, L_J likewise   (64x32 each, re-derived per tile from the stored pivot block)
//   C[I, J] -= L_I L_J'
// The tj == 0 tiles store L_I into the upper triangle (L[i][k] at row k, column i).  The first `npiv`
// workgroups of the launch are pivot workgroups (pivot_path): one per front that has a next panel, they do only
// what the NEXT launch waits for -- the 32 panel rows, the 32x32 corner of the update and its factorisation.
// Every row x of X L11' = A (A, X: 1 x 32) is solved by right-looking substitution against the pivot block (LDS, reciprocal
// diagonal), software pipelined: the columns of L11 are requested two steps ahead of their use.
// The substitution runs with FOUR lanes per row (a quad: lanes 4 r .. 4 r + 3 of a wave, lane c4 owning the entries
// m = c4, c4 + 4, .. of the row in f[0..8)): the 496 multiply-adds of a row are split four ways and the solved entry
// x_j travels from its owner to the other three lanes with a quad-broadcast DPP move (one VALU pass, no LDS), so a wave
// solves 16 rows and all four waves of a workgroup work -- with one lane per row a wave spent ~3 us on its 64 rows while
// two or three waves idled.  Operation order per entry is that of the one-lane-per-row loop (for j: x_j = f_j / L_jj;
// f_m -= x_j L_mj, m > j): bitwise the same result.
// Lq: the pivot block in "quad" order, Lq[32 j + 8 (m & 3) + (m >> 2)] = L[m][j] (reciprocal diagonal), so that the
// entries of column j a lane needs are contiguous (b128 LDS reads).
__device__ __forceinline__ int lq_index(int m, int j) { return 32 * j + 8 * (m & 3) + (m >> 2); }

template <int OWNER>
__device__ __forceinline__ double quad_bcast(double v) {
  constexpr int ctrl = OWNER * 0x55;      // quad_perm: every lane of the quad reads lane OWNER
  const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), ctrl, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), ctrl, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}

// PACKED: the factor is read straight from the row-major packed lower triangle (front_leaf keeps no second copy: LDS is what
// limits its four workgroups per CU); entries above the diagonal are never used (their slot reads a valid, ignored word)
template <int J, bool PACKED>
__device__ __forceinline__ void trsm_quad_load(double (&col)[PB / 4], double& diag, const double* Lq, int c4) {
  if constexpr (J < PB) {
    if (PACKED) {
#pragma unroll
      for (int k = J >> 2; k < PB / 4; ++k) col[k] = Lq[lo_packed(c4 + 4 * k, J)];
      diag = Lq[lo_packed(J, J)];
    } else {
#pragma unroll
      for (int k = J >> 2; k < PB / 4; ++k) col[k] = Lq[32 * J + 8 * c4 + k];      // this lane's entries of column J: L[c4 + 4 k][J]
      diag = Lq[32 * J + 8 * (J & 3) + (J >> 2)];                                   // 1 / L[J][J]
    }
  }
}

// software pipeline: the entries of column J + 2 are requested before column J is consumed (two columns in flight cover
// the LDS latency of the short steps); the empty asm keeps the compiler from hoisting ALL columns (it did: 372 VGPRs)
// or sinking the loads next to their uses.
// NB row blocks (16 rows each, their own f / x registers) are solved side by side against the same columns: a single block
// is bound by the latency of its dependent chain (scale, broadcast, update: ~125 cycles per step against ~85 of issue),
// so a second and a third block ride in the gaps -- three blocks take about as long as one.
// The solved entry goes to x[kk] of its owner by a multiply-add with the lane's 0 / 1 owner flag (one instruction; the
// select it replaces took two, and two more for the lane's own entry of the column, which (not PACKED) needs none: left
// of the diagonal the stored factor holds zeros, and the owner's f[kk] -- "updated" with the reciprocal pivot -- is dead
// from here on).
template <int J, int NB, bool EXIT, bool PACKED>
__device__ __forceinline__ void trsm_quad_step(double (&f)[NB][PB / 4], double (&x)[NB][PB / 4], const double (&e)[4], const double* Lq,
                                               int c4, int kw, const double (&col)[PB / 4], double diag, const double (&col1)[PB / 4],
                                               double diag1) {
  if constexpr (J < PB) {
    constexpr int owner = J & 3, kk = J >> 2;
    if (EXIT && (J & 7) == 0 && J >= kw) {      // identity padding (workgroup-uniform): the remaining entries pass through
#pragma unroll
      for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int k = kk; k < PB / 4; ++k) x[b][k] = f[b][k];
      return;
    }
    double col2[PB / 4], diag2 = 0.0;
    trsm_quad_load<J + 2, PACKED>(col2, diag2, Lq, c4);
    asm volatile("" ::: "memory");
    const double lkk = PACKED ? ((c4 > owner) ? col[kk] : 0.0) : col[kk];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const double fj = quad_bcast<owner>(f[b][kk] * diag);
      x[b][kk] = fma(fj, e[owner], x[b][kk]);
      f[b][kk] = fma(-fj, lkk, f[b][kk]);
#pragma unroll
      for (int k = kk + 1; k < PB / 4; ++k) f[b][k] = fma(-fj, col[k], f[b][k]);
    }
    trsm_quad_step<J + 1, NB, EXIT, PACKED>(f, x, e, Lq, c4, kw, col1, diag1, col2, diag2);
  }
}

// f (in): this lane's entries m = c4, c4 + 4, .. of NB rows; (out): the solved entries
template <int NB, bool EXIT = false, bool PACKED = false>
__device__ __forceinline__ void trsm_quad_n(double (&f)[NB][PB / 4], const double* Lq, int c4, int kw = PB) {
  double col[PB / 4], diag = 0.0, col1[PB / 4], diag1 = 0.0, x[NB][PB / 4], e[4];
#pragma unroll
  for (int o = 0; o < 4; ++o) e[o] = (c4 == o) ? 1.0 : 0.0;
#pragma unroll
  for (int b = 0; b < NB; ++b)
#pragma unroll
    for (int k = 0; k < PB / 4; ++k) x[b][k] = 0.0;
  trsm_quad_load<0, PACKED>(col, diag, Lq, c4);
  trsm_quad_load<1, PACKED>(col1, diag1, Lq, c4);
  trsm_quad_step<0, NB, EXIT, PACKED>(f, x, e, Lq, c4, kw, col, diag, col1, diag1);
#pragma unroll
  for (int b = 0; b < NB; ++b)
#pragma unroll
    for (int k = 0; k < PB / 4; ++k) f[b][k] = x[b][k];
}

template <bool EXIT = false, bool PACKED = false>
__device__ __forceinline__ void trsm_quad(double (&f)[PB / 4], const double* Lq, int c4, int kw = PB) {
  double g[1][PB / 4];
#pragma unroll
  for (int k = 0; k < PB / 4; ++k) g[0][k] = f[k];
  trsm_quad_n<1, EXIT, PACKED>(g, Lq, c4, kw);
#pragma unroll
  for (int k = 0; k < PB / 4; ++k) f[k] = g[0][k];
}

// X L11' = A for the 64 rows of a staged panel block AT (AT[q * TP + r] = (row r, panel column q)), in place, by all four
// waves: wave w takes rows 16 w .. 16 w + 15, four lanes per row.
template <bool EXIT = false, bool PACKED = false>
__device__ __forceinline__ void trsm_block64(double* AT, int TP, const double* Lq, int kw = PB) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, r = 16 * w + (lane >> 2), c4 = lane & 3;
  double f[PB / 4];
#pragma unroll
  for (int k = 0; k < PB / 4; ++k) f[k] = AT[(c4 + 4 * k) * TP + r];
  trsm_quad<EXIT, PACKED>(f, Lq, c4, kw);
#pragma unroll
  for (int k = 0; k < PB / 4; ++k) AT[(c4 + 4 * k) * TP + r] = f[k];
}

// the same for the two panel blocks of an off-diagonal tile at once (rows 16 w .. of both, side by side in every wave)
template <bool EXIT = false, bool PACKED = false>
__device__ __forceinline__ void trsm_block64_pair(double* ATI, double* ATJ, int TP, const double* Lq, int kw = PB) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, r = 16 * w + (lane >> 2), c4 = lane & 3;
  double f[2][PB / 4];
#pragma unroll
  for (int k = 0; k < PB / 4; ++k) {
    f[0][k] = ATI[(c4 + 4 * k) * TP + r];
    f[1][k] = ATJ[(c4 + 4 * k) * TP + r];
  }
  trsm_quad_n<2, EXIT, PACKED>(f, Lq, c4, kw);
#pragma unroll
  for (int k = 0; k < PB / 4; ++k) {
    ATI[(c4 + 4 * k) * TP + r] = f[0][k];
    ATJ[(c4 + 4 * k) * TP + r] = f[1][k];
  }
}

// Critical path of a panel step, run by a dedicated workgroup per front: rows k1..k1+31 of the panel are solved
// against the current pivot block, the 32x32 corner C[k1:k1+32, k1:k1+32] gets its rank-32 update, and the result
// -- the next pivot block -- is factored and published.  The regular tile (0,0) of the same launch handles the
// rest of its 64x64 tile (and the mirrored copy of these rows) but leaves the corner alone; nobody reads the
// corner again once its factor is stored.
__device__ inline void pivot_path(const StepTile& t, int p, double* sh, double* fronts,
                                  const double* __restrict__ linv_ro, double* linv, int* fail, long long* prof) {
  double* Lc = sh;                     // column-major pivot block of THIS panel
  double* P = sh + PB * PB;            // solved panel rows, P[r * LP + q]
  double* D = P + PB * LP;             // updated corner = next pivot block
  double* Lo = D + PB * LP;            // scratch of the factorisation
  const int nf = t.nf, ld = nf + 1, k0 = p * PB, kw = min(PB, t.ns - k0), k1 = k0 + kw, kw2 = min(PB, t.ns - k1);
  double* F = fronts + t.off;
  const int tid = threadIdx.x;
  for (int idx = tid; idx < PB * PB; idx += TB) Lc[lq_index(idx % PB, idx / PB)] = linv_ro[t.loff + PB * PB + idx];      // quad order
  // the corner as three 16x16 blocks of the lower triangle, one per wave (wave 3 idles), in the result layout of
  // v_mfma_f64_16x16x4_f64 with m = corner column, n = corner row: lane (li, lk) holds (row 16 bi + li, column
  // 16 bj + lk + 4 reg), so the loads run along the rows of the front
  typedef double v4f64 __attribute__((ext_vector_type(4)));
  const int lane = tid & 63, w = tid >> 6, li = lane & 15, lk = lane >> 4;
  const int bi = (w + 1) >> 1, bj = w >> 1;      // waves 0, 1, 2 -> blocks (0,0), (1,0), (1,1)
  const int ci = 16 * bi + li;
  v4f64 acc = v4f64{0.0, 0.0, 0.0, 0.0};
  if (w < 3) {
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const int j = 16 * bj + lk + 4 * reg;
      acc[reg] = (j <= ci && ci < kw2) ? F[(long long)ld * (k1 + j) + k1 + ci] : 0.0;
    }
  }
  // the 32 panel rows below the pivot block: waves 0 and 1, 16 rows each, four lanes per row (trsm_quad)
  const int pr = 16 * w + (lane >> 2), c4 = lane & 3;
  double f[PB / 4];
  if (w < 2) {
#pragma unroll
    for (int k = 0; k < PB / 4; ++k) {
      const int m = c4 + 4 * k;
      f[k] = (m < kw && pr < kw2) ? F[(long long)ld * (k0 + m) + k1 + pr] : 0.0;
    }
  }
  __syncthreads();
  STAMP(1);
  if (w < 2) {      // wave-uniform
    trsm_quad(f, Lc, c4);
#pragma unroll
    for (int k = 0; k < PB / 4; ++k) P[pr * LP + c4 + 4 * k] = f[k];
  }
  __syncthreads();
  STAMP(2);
  if (w < 3) {      // wave-uniform: C - P_i P_j' on the matrix cores (negated row operand, C as accumulator input)
    const double* PI = P + ci * LP + lk;
    const double* PJ = P + (16 * bj + li) * LP + lk;
#pragma unroll
    for (int ks = 0; ks < PB / 4; ++ks) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(PJ[4 * ks], -PI[4 * ks], acc, 0, 0, 0);
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) D[ci * LP + 16 * bj + lk + 4 * reg] = acc[reg];      // entries right of the diagonal: don't care
  }
  STAMP(3);
  __syncthreads();
  STAMP(4);
  STAMP(5);
  factor_diag_block(D, kw2, Lo, linv + t.loff + 2 * PB * PB, fail, prof);
  STAMP(7);
}


__global__ __launch_bounds__(TB) __attribute__((amdgpu_waves_per_eu(1, 3))) void front_step_kernel(const StepTile* __restrict__ tiles, int p, int npiv, double* fronts,
                                                         const double* __restrict__ linv_ro, double* linv, int* fail,
                                                         long long* prof) {
  // panel rows of the tile, transposed: AT[q * TP + r] = (row r, panel column q): unit stride over the rows for the
  // staging, the substitution (lane = row) and the MFMA operand reads (16 rows x 4 panel columns per read).
  constexpr int TP = TS + 8;      // row stride of the staged panel blocks: the four k-groups of an MFMA operand read land on disjoint banks
  __shared__ __attribute__((aligned(32))) double sh[2 * PB * TP + PB * PB];
  double* ATI = sh;
  double* ATJ = sh + PB * TP;
  double* Lc = sh + 2 * PB * TP;      // column-major pivot block: Lc[32 j + m] = L[m][j]
  STAMP(0);
  const StepTile t = tiles[blockIdx.x];
  if ((int)blockIdx.x < npiv) {      // workgroup-uniform: the dedicated pivot workgroup of one front
    pivot_path(t, p, sh, fronts, linv_ro, linv, fail, prof);
    return;
  }
  const int nf = t.nf, ld = nf + 1, k0 = p * PB, kw = min(PB, t.ns - k0), k1 = k0 + kw;
  double* F = fronts + t.off;
  const int r0 = k1 + TS * t.ti, c0 = k1 + TS * t.tj;
  const bool diag = (t.ti == t.tj);
  const int tid = threadIdx.x;
  for (int idx = tid; idx < PB * PB; idx += TB) Lc[lq_index(idx % PB, idx / PB)] = linv_ro[t.loff + PB * PB + idx];      // quad order
  // panel rows of the tile, transposed and padded: AT[q * TP + r] = (row r, panel column q)
  for (int idx = tid; idx < TS * PB; idx += TB) {
    const int r = idx % TS, q = idx / TS;
    ATI[q * TP + r] = (q < kw && r0 + r <= nf) ? F[(long long)ld * (k0 + q) + r0 + r] : 0.0;
    if (!diag) ATJ[q * TP + r] = (q < kw && c0 + r <= nf) ? F[(long long)ld * (k0 + q) + c0 + r] : 0.0;
  }
  // the C tile is requested now, in the MFMA result layout (see below), so that its global latency overlaps the substitution
  const int kc = (t.ti == 0 && t.tj == 0) ? min(t.ns, k1 + PB) : 0;      // rows/columns < kc: the next pivot block
  const int lane = tid & 63, w = tid >> 6, li = lane & 15, lk = lane >> 4;
  const int i = r0 + 16 * w + li;
  double c[4][4];
#pragma unroll
  for (int bj = 0; bj < 4; ++bj)
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const int j = c0 + 16 * bj + lk + 4 * reg;
      c[bj][reg] = (i <= nf && j < nf && i >= j && !(i < kc && j < kc)) ? F[(long long)ld * j + i] : 0.0;
    }
  __syncthreads();
  STAMP(1);
  // X L11' = A for the rows of I (and of J off the diagonal) by all four waves, four lanes per row (trsm_block64)
  if (diag) trsm_block64(ATI, TP, Lc);
  else trsm_block64_pair(ATI, ATJ, TP, Lc);
  __syncthreads();
  // finished rows of L go to the mirrored (upper) half (row r0 + r, 32 consecutive entries), stored by the LAST tile of the
  // row block -- its diagonal tile, which stages one panel block where an off-diagonal tile stages and solves two and is what
  // the launch waits for (the tj == 0 tiles did this before: 1.5 us on the slowest workgroups)
  if (t.tj == min((int)t.ti, max(1, (nf - k1 + TS - 1) / TS) - 1)) {
    for (int idx = tid; idx < TS * PB; idx += TB) {
      const int r = idx / PB, m = idx % PB;
      if (r0 + r <= nf && m < kw) F[(long long)ld * (r0 + r) + k0 + m] = ATI[m * TP + r];
    }
  }
  STAMP(2);
  // Rank-32 update C[I, J] -= L_I L_J' on the matrix cores: v_mfma_f64_16x16x4_f64, D[m][n] += A[m][k] B[k][n] with
  // m = tile column (L_J), n = tile row (L_I), so that the lanes of a result register run along the rows of the
  // front (unit stride in HBM).  Wave w owns tile rows 16 w .. 16 w + 15 and all four 16-column blocks: lane
  // (li = lane & 15, lk = lane >> 4) holds C[16 w + li][16 bj + lk + 4 reg] in acc[bj][reg]; per k-step of 4 panel
  // columns it reads ONE L_I entry and four L_J entries from LDS (a quarter of the operand traffic of the 4x4
  // register micro-tiles).  The 32x32 corner that becomes the next pivot block belongs to the pivot workgroup.
  typedef double v4f64 __attribute__((ext_vector_type(4)));
  const double* LI = ATI + 16 * w + li + lk * TP;
  const double* LJ = (diag ? ATI : ATJ) + li + lk * TP;
  v4f64 acc[4];      // starts as C; the negated L_I operand makes the matrix core return C - L_I L_J'
#pragma unroll
  for (int bj = 0; bj < 4; ++bj) acc[bj] = v4f64{c[bj][0], c[bj][1], c[bj][2], c[bj][3]};
#pragma unroll
  for (int ks = 0; ks < PB / 4; ++ks) {
    const double bv = -LI[4 * ks * TP];
#pragma unroll
    for (int bj = 0; bj < 4; ++bj)
      acc[bj] = __builtin_amdgcn_mfma_f64_16x16x4f64(LJ[4 * ks * TP + 16 * bj], bv, acc[bj], 0, 0, 0);
  }
#pragma unroll
  for (int bj = 0; bj < 4; ++bj)
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const int j = c0 + 16 * bj + lk + 4 * reg;
      if (i <= nf && j < nf && i >= j && !(i < kc && j < kc)) F[(long long)ld * j + i] = acc[bj][reg];
    }
  STAMP(3);
  STAMP(7);
}

// TWO panels per launch (columns k0..k0+63 of every front of the height that still has them): halves the launches of the
// multi-panel heights and the passes over the trailing matrix (what bounds the large 3-D fronts: a rank-64 instead of two
// rank-32 updates per C tile).  The first pivot block L11 comes from the previous launch as before.  The second one cannot
// (it needs the first panel's update), so EVERY workgroup re-derives it: L21 = A21 L11^-T (32 rows), A22 - L21 L21' on the
// matrix cores, factor -- pivot_path's work, 5 us of redundant flops instead of a launch boundary, as front_single does --
// and while wave 0 factors, waves 1..3 already solve the tile's panel blocks against L11 and subtract X1 L21' from their
// second halves.  Then X2 = (.) L22^-T, the mirrored rows of L, and one rank-64 update of the C tile.  Tile (0,0) of every
// front publishes L21 and the second pivot block; the pivot workgroups (first npiv) do the same for rows k2..k2+31 only
// and factor the corner that becomes the pivot block of the NEXT launch.
// LDS (dynamic, 115 KB): L11 and L22 in quad order, L21, the corner, the factor scratch, two 64 x 64 panel blocks.
constexpr int kStep2Lds = (2 * PB * PB + 3 * PB * LP + 2 * 2 * PB * (TS + 8)) * (int)sizeof(double);
// The same two panels as TWO launches, for the heights whose tile count is beyond what one workgroup per CU handles in a
// round (fem2d L >= 8, the upper heights of fem3d): there the one-panel kernel was used, which re-reads and re-writes the
// whole trailing matrix per 32 columns.
//   MODE 1, "panel": one workgroup per block of trailing rows.  Everything of the fused kernel up to the solved panel blocks
//     X = [X1 X2] of its rows, which go to the mirrored rows of L AND back into the panel's own (column-major) place, dead
//     from here on in the forward sweep.  The first 32 trailing rows belong to the front's pivot workgroup (first in the
//     launch: it publishes L21 / L22 and factors the next pivot block from those rows, the chain's critical path), the
//     64-row blocks start behind them -- nobody reads panel rows another workgroup overwrites.  78 KB of LDS (no second
//     panel block): two per CU.
//   MODE 2, "update": the fused kernel's tiles with everything between staging and the rank-64 update removed -- the staged
//     panel blocks ARE the solved ones.  74 KB of LDS: two per CU.
// Operation order per entry is that of the fused kernel (and of two rank-32 steps): bitwise the same factor.
constexpr int kPanel2Lds = (2 * PB * PB + 3 * PB * LP + 2 * PB * (TS + 8)) * (int)sizeof(double);
constexpr int kUpdate2Lds = (2 * 2 * PB * (TS + 8)) * (int)sizeof(double);

template <int MODE>
__device__ __forceinline__ void front_step2_body(
    const StepTile* __restrict__ tiles, int p, int npiv, double* fronts, const double* __restrict__ linv_ro, double* linv, int* fail,
    long long* prof) {
  constexpr int TP = TS + 8;
  extern __shared__ __attribute__((aligned(32))) double sh2[];
  double* Lc1 = sh2;                    // L11, quad order
  double* Lc2 = Lc1 + PB * PB;          // L22, quad order
  double* P = Lc2 + PB * PB;            // L21: P[r * LP + m]
  double* D = P + PB * LP;              // corner A22 - L21 L21' (pivot workgroups: later the next launch's pivot block)
  double* Lo = D + PB * LP;             // factor scratch / L22 row-major
  double* ATI = MODE == 2 ? sh2 : Lo + PB * LP;      // AT[q * TP + r]: q < 32 first panel, q >= 32 second panel
  double* ATJ = ATI + 2 * PB * TP;                   // (MODE 1: never touched, not allocated)
  STAMP(0);
  const StepTile t = tiles[blockIdx.x];
  const bool is_piv = MODE != 2 && (int)blockIdx.x < npiv;      // workgroup-uniform
  const int nf = t.nf, ld = nf + 1, k0 = p * PB, k1 = min(t.ns, k0 + PB), k2 = min(t.ns, k1 + PB), kw = k1 - k0, kw2 = k2 - k1;
  const int kw3 = min(PB, t.ns - k2);              // pivot workgroups: width of the next pivot block
  double* F = fronts + t.off;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, li = lane & 15, lk = lane >> 4;
  const int roff = MODE == 1 ? PB : 0;             // panel launch: the 64-row blocks start behind the pivot workgroup's 32 rows
  const int r0 = is_piv ? k2 : k2 + roff + TS * t.ti, c0 = is_piv ? k2 : k2 + TS * t.tj;
  const bool diag = is_piv || (t.ti == t.tj);
  // pivot workgroups stage only the rows of the next pivot block (panel launch: all of their 32 rows)
  const int rows_here = is_piv ? (MODE == 1 ? PB : kw3) : TS;
  typedef double v4f64 __attribute__((ext_vector_type(4)));
  if (MODE != 2)
    for (int idx = tid; idx < PB * PB; idx += TB) Lc1[lq_index(idx % PB, idx / PB)] = linv_ro[t.loff + PB * PB + idx];
  // second pivot block: its 32 rows of the first panel (waves 0, 1: four lanes per row) and its lower triangle as three
  // 16x16 blocks in the MFMA result layout (waves 0..2) -- see pivot_path
  const int bi = (w + 1) >> 1, bj3 = w >> 1, ci = 16 * bi + li;
  v4f64 accD = v4f64{0.0, 0.0, 0.0, 0.0};
  if (MODE != 2 && w < 3) {
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const int j = 16 * bj3 + lk + 4 * reg;
      accD[reg] = (j <= ci && ci < kw2) ? F[(long long)ld * (k1 + j) + k1 + ci] : 0.0;
    }
  }
  const int pr = 16 * w + (lane >> 2), c4 = lane & 3;
  double f[PB / 4];
  if (MODE != 2 && w < 2) {
#pragma unroll
    for (int k = 0; k < PB / 4; ++k) {
      const int m = c4 + 4 * k;
      f[k] = (m < kw && pr < kw2) ? F[(long long)ld * (k0 + m) + k1 + pr] : 0.0;
    }
  }
  // the two-panel blocks of the tile's rows (and columns off the diagonal): thread = one row, every fourth column; ALL loads
  // are issued before the first LDS store (the store-as-you-go loop serialised into ~4 memory round trips, 7.4 us)
  const int sr = tid & (TS - 1), sq0 = tid >> 6;
  double vI[2 * PB / 4], vJ[2 * PB / 4];
  {
    const bool okI = sr < rows_here && r0 + sr <= nf, okJ = !diag && c0 + sr <= nf;
#pragma unroll
    for (int u = 0; u < 2 * PB / 4; ++u) {
      const int q = sq0 + 4 * u;
      const int col = (q < PB) ? k0 + q : k1 + q - PB;
      const bool okc = (q < PB) ? (q < kw) : (q - PB < kw2);
      const double* src = F + (long long)ld * (okc ? col : k0);      // clamped to a valid column when there is none
      vI[u] = okI ? src[r0 + sr] : 0.0;
      vJ[u] = okJ ? src[c0 + sr] : 0.0;
      if (!okc) vI[u] = vJ[u] = 0.0;
    }
  }
  // C tile (pivot workgroups: the corner of the next pivot block, on waves 0..2), requested now
  const int kc = (!is_piv && t.ti == 0 && t.tj == 0) ? min(t.ns, k2 + PB) : 0;
  const int i = r0 + 16 * w + li;
  double c[4][4];
  v4f64 accN = v4f64{0.0, 0.0, 0.0, 0.0};
  if (is_piv) {
    if (w < 3) {
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int j = 16 * bj3 + lk + 4 * reg;
        accN[reg] = (j <= ci && ci < kw3) ? F[(long long)ld * (k2 + j) + k2 + ci] : 0.0;
      }
    }
  }
  if (MODE != 1 && !is_piv) {
#pragma unroll
    for (int bj = 0; bj < 4; ++bj)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int j = c0 + 16 * bj + lk + 4 * reg;
        c[bj][reg] = (i <= nf && j < nf && i >= j && !(i < kc && j < kc)) ? F[(long long)ld * j + i] : 0.0;
      }
  }
#pragma unroll
  for (int u = 0; u < 2 * PB / 4; ++u) {
    ATI[(sq0 + 4 * u) * TP + sr] = vI[u];
    if (!diag) ATJ[(sq0 + 4 * u) * TP + sr] = vJ[u];
  }
  __syncthreads();
  STAMP(1);
  if (MODE != 2) {      // (the update launch's staged blocks are the solved ones)
  if (w < 2) {      // L21 = A21 L11^-T
    trsm_quad(f, Lc1, c4);
#pragma unroll
    for (int k = 0; k < PB / 4; ++k) P[pr * LP + c4 + 4 * k] = f[k];
  }
  __syncthreads();
  if (w < 3) {      // corner - L21 L21'
    const double* PI = P + ci * LP + lk;
    const double* PJ = P + (16 * bj3 + li) * LP + lk;
#pragma unroll
    for (int ks = 0; ks < PB / 4; ++ks) accD = __builtin_amdgcn_mfma_f64_16x16x4f64(PJ[4 * ks], -PI[4 * ks], accD, 0, 0, 0);
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) D[ci * LP + 16 * bj3 + lk + 4 * reg] = accD[reg];
  }
  __syncthreads();
  STAMP(2);
  if (w == 0) {
    factor_block_wave0(D, kw2, Lo, fail);      // L22
  } else {
    // 16-row blocks of the panel blocks: X1 = A1 L11^-T in place, then A2 -= X1 L21' (both wave-local).  Wave w takes
    // the blocks w - 1, w + 2, w + 5 (of 2 / 4 / 8: pivot workgroup, diagonal tile, off-diagonal tile) and solves them
    // side by side (trsm_quad_n: three blocks take about as long as one, and wave 0 is busy with the factor that long)
    const int nblk = is_piv ? 2 : (diag ? 4 : 8);
    auto block_at = [&](int b) { return ((b < 4) ? ATI : ATJ) + 16 * (b & 3); };      // rows of block b: + r
    auto update2 = [&](double* AB) {      // A2 -= X1 L21' for the 16 rows at AB
      v4f64 a2[2];
#pragma unroll
      for (int bj = 0; bj < 2; ++bj)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) a2[bj][reg] = AB[(PB + 16 * bj + lk + 4 * reg) * TP + li];
#pragma unroll
      for (int ks = 0; ks < PB / 4; ++ks) {
        const double bv = -AB[(4 * ks + lk) * TP + li];
#pragma unroll
        for (int bj = 0; bj < 2; ++bj)
          a2[bj] = __builtin_amdgcn_mfma_f64_16x16x4f64(P[(16 * bj + li) * LP + 4 * ks + lk], bv, a2[bj], 0, 0, 0);
      }
#pragma unroll
      for (int bj = 0; bj < 2; ++bj)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) AB[(PB + 16 * bj + lk + 4 * reg) * TP + li] = a2[bj][reg];
    };
    const int b0 = w - 1, r = lane >> 2;
    if (b0 + 3 >= nblk) {      // one block (wave-uniform; the pivot workgroups, whose factor is the critical path of the chain)
      if (b0 < nblk) {
        double* AB = block_at(b0);
        double g[PB / 4];
#pragma unroll
        for (int k = 0; k < PB / 4; ++k) g[k] = AB[(c4 + 4 * k) * TP + r];
        trsm_quad(g, Lc1, c4);
#pragma unroll
        for (int k = 0; k < PB / 4; ++k) AB[(c4 + 4 * k) * TP + r] = g[k];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        update2(AB);
      }
    } else {                   // two or three blocks: a missing third one is solved as zeros and not stored
      const bool third = b0 + 6 < nblk;
      double* AB[3] = {block_at(b0), block_at(b0 + 3), block_at(third ? b0 + 6 : b0)};
      double g[3][PB / 4];
#pragma unroll
      for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int k = 0; k < PB / 4; ++k) g[q][k] = (q < 2 || third) ? AB[q][(c4 + 4 * k) * TP + r] : 0.0;
      trsm_quad_n<3>(g, Lc1, c4);
#pragma unroll
      for (int q = 0; q < 3; ++q)
        if (q < 2 || third) {
#pragma unroll
          for (int k = 0; k < PB / 4; ++k) AB[q][(c4 + 4 * k) * TP + r] = g[q][k];
        }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      update2(AB[0]);
      update2(AB[1]);
      if (third) update2(AB[2]);
    }
  }
  __syncthreads();
  STAMP(3);
  for (int idx = tid; idx < PB * PB; idx += TB) Lc2[lq_index(idx % PB, idx / PB)] = Lo[(idx % PB) * LP + idx / PB];
  if ((MODE == 1 ? is_piv : (!is_piv && t.ti == 0 && t.tj == 0)) && kw2 > 0) {      // this front's publisher of the second pivot block and of the rows of L21 (if it has a second panel)
    double* lp = linv + t.loff + 2 * PB * PB;
    for (int idx = tid; idx < PB * PB; idx += TB) {
      const int r = idx / PB, m = idx % PB;
      lp[idx] = Lo[r * LP + m];
      lp[PB * PB + idx] = Lo[m * LP + r];
      if (r < kw2 && m < kw) F[(long long)ld * (k1 + r) + k0 + m] = P[r * LP + m];
    }
  }
  __syncthreads();
  if (diag) trsm_block64(ATI + PB * TP, TP, Lc2);      // X2 = (A2 - X1 L21') L22^-T
  else trsm_block64_pair(ATI + PB * TP, ATJ + PB * TP, TP, Lc2);
  __syncthreads();
  STAMP(4);
  if (MODE == 1) {
    // the solved rows go to the mirrored half (the backward sweep's rows of L) and back into the panel itself, where the
    // update launch stages them from like the fused kernel stages the unsolved ones
    for (int idx = tid; idx < TS * 2 * PB; idx += TB) {
      const int r = idx / (2 * PB), q = idx % (2 * PB);
      const int col = (q < PB) ? k0 + q : k1 + q - PB;
      const bool okc = (q < PB) ? (q < kw) : (q - PB < kw2);
      if (r < rows_here && r0 + r <= nf && okc) F[(long long)ld * (r0 + r) + col] = ATI[q * TP + r];
    }
    for (int idx = tid; idx < TS * 2 * PB; idx += TB) {
      const int r = idx % TS, q = idx / TS;
      const int col = (q < PB) ? k0 + q : k1 + q - PB;
      const bool okc = (q < PB) ? (q < kw) : (q - PB < kw2);
      if (r < rows_here && r0 + r <= nf && okc) F[(long long)ld * col + r0 + r] = ATI[q * TP + r];
    }
    if (!(is_piv && k2 < t.ns)) return;      // workgroup-uniform: only a pivot workgroup with a next pivot block goes on
  }
  if (is_piv) {
    // next pivot block: corner - [X1 X2] [X1 X2]' over its 32 rows, then the factor the NEXT launch starts from
    if (w < 3) {
      const double* PI = ATI + ci + lk * TP;
      const double* PJ = ATI + 16 * bj3 + li + lk * TP;
#pragma unroll
      for (int ks = 0; ks < 2 * PB / 4; ++ks) accN = __builtin_amdgcn_mfma_f64_16x16x4f64(PJ[4 * ks * TP], -PI[4 * ks * TP], accN, 0, 0, 0);
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) D[ci * LP + 16 * bj3 + lk + 4 * reg] = accN[reg];
    }
    __syncthreads();
    STAMP(5);
    factor_diag_block(D, kw3, Lo, linv + t.loff + 4 * PB * PB, fail, prof);
    STAMP(7);
    return;
  }
  }      // MODE != 2
  if (MODE == 0 && t.tj == min((int)t.ti, max(1, (nf - k2 + TS - 1) / TS) - 1)) {      // finished rows of L (both panels) go to the mirrored half: the row block's last (diagonal) tile, see front_step
    for (int idx = tid; idx < TS * 2 * PB; idx += TB) {
      const int r = idx / (2 * PB), q = idx % (2 * PB);
      const int col = (q < PB) ? k0 + q : k1 + q - PB;
      const bool okc = (q < PB) ? (q < kw) : (q - PB < kw2);
      if (r0 + r <= nf && okc) F[(long long)ld * (r0 + r) + col] = ATI[q * TP + r];
    }
  }
  STAMP(5);
  // rank-64 update of the C tile (layout and operand order of front_step)
  const double* LI = ATI + 16 * w + li + lk * TP;
  const double* LJ = (diag ? ATI : ATJ) + li + lk * TP;
  v4f64 acc[4];
#pragma unroll
  for (int bj = 0; bj < 4; ++bj) acc[bj] = v4f64{c[bj][0], c[bj][1], c[bj][2], c[bj][3]};
#pragma unroll
  for (int ks = 0; ks < 2 * PB / 4; ++ks) {
    const double bv = -LI[4 * ks * TP];
#pragma unroll
    for (int bj = 0; bj < 4; ++bj)
      acc[bj] = __builtin_amdgcn_mfma_f64_16x16x4f64(LJ[4 * ks * TP + 16 * bj], bv, acc[bj], 0, 0, 0);
  }
  STAMP(6);
#pragma unroll
  for (int bj = 0; bj < 4; ++bj)
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const int j = c0 + 16 * bj + lk + 4 * reg;
      if (i <= nf && j < nf && i >= j && !(i < kc && j < kc)) F[(long long)ld * j + i] = acc[bj][reg];
    }
  STAMP(7);
}

__global__ __launch_bounds__(TB) __attribute__((amdgpu_waves_per_eu(1, 1))) void front_step2_kernel(
    const StepTile* __restrict__ tiles, int p, int npiv, double* fronts, const double* __restrict__ linv_ro, double* linv, int* fail,
    long long* prof) {
  front_step2_body<0>(tiles, p, npiv, fronts, linv_ro, linv, fail, prof);
}

__global__ __launch_bounds__(TB) __attribute__((amdgpu_waves_per_eu(2, 2))) void front_panel2_kernel(
    const StepTile* __restrict__ tiles, int p, int npiv, double* fronts, const double* __restrict__ linv_ro, double* linv, int* fail,
    long long* prof) {
  front_step2_body<1>(tiles, p, npiv, fronts, linv_ro, linv, fail, prof);
}

__global__ __launch_bounds__(TB) __attribute__((amdgpu_waves_per_eu(2, 2))) void front_update2_kernel(
    const StepTile* __restrict__ tiles, int p, int npiv, double* fronts, const double* __restrict__ linv_ro, double* linv, int* fail,
    long long* prof) {
  front_step2_body<2>(tiles, p, npiv, fronts, linv_ro, linv, fail, prof);
}

// Heights whose fronts all have a single panel (ns <= 32; at fem2d L=7 five of the eleven heights): front_start and
// the one front_step collapse into ONE launch without any dependency between its workgroups.  Workgroup = 64x64
// tile of the trailing matrix, as in front_step, but it GATHERS what it needs instead of reading an assembled front:
// the pivot block (every tile re-derives and factors it -- 5 us of redundant work instead of a launch), its two
// 64x32 panel blocks, and its C tile, each as child0 + child1 contribution through the inverse index maps, plus
// the assembled matrix entries and the right-hand side (which only live in pivot columns).  Tile (0,0) publishes
// the pivot block; the tj == 0 tiles the mirrored rows of L.
// NARROW: launches whose fronts have at most 8 pivots skip the identity padding in the factor, the substitution,
// the update and the gather of the panel columns.
// DENSE (launches with more tiles than two per CU can hold at once, 512: fem2d L >= 8, where these launches run up to 16 rounds
// of workgroups and are bound by how many of them a CU holds): the factor stays in its packed row-major form and the
// substitutions read it there (as front_leaf does) instead of from a second copy in quad order: 51 instead of 63 KB of
// LDS, three workgroups per CU instead of two.  Same operations in the same order: bitwise the same result.
template <bool NARROW, bool DENSE>
__device__ __forceinline__ void front_single_body(
    const SingleTile* __restrict__ tiles, const int* __restrict__ pinv,
    const int* __restrict__ asm_src, const int* __restrict__ asm_pos, const double* __restrict__ vals,
    const int* __restrict__ perm, const double* __restrict__ b, const double* __restrict__ fronts_ro, double* fronts,
    double* linv, int* fail, long long* prof) {
  constexpr int TP = TS + 8;      // row stride of the staged panel blocks (as in front_step)
  constexpr int kFactorLds = DENSE ? PB * LP + PB * (PB + 1) / 2 : PB * PB + 2 * PB * LP;      // D | Lo (packed)  or  Lc | D | Lo
  __shared__ __attribute__((aligned(32))) double sh[2 * PB * TP + kFactorLds];
  __shared__ int rowI[2][TS], rowJ[2][TS], piv[2][PB];
  double* ATI = sh;
  double* ATJ = sh + PB * TP;
  double* Lc = sh + 2 * PB * TP;                       // (not DENSE) the factor in quad order
  double* D = DENSE ? sh + 2 * PB * TP : Lc + PB * PB;
  double* Lo = D + PB * LP;
  STAMP(0);
  const SingleTile t = tiles[blockIdx.x];      // carries everything of the node and its children: no dependent loads
  const int nf = t.nf, ld = nf + 1, ns = t.ns, kw = ns, k1 = ns;      // ns <= 32: one panel, k0 = 0
  double* F = fronts + t.off;
  const int r0 = k1 + TS * t.ti, c0 = k1 + TS * t.tj;
  const bool diag = (t.ti == t.tj);
  const int tid = threadIdx.x;
  const int lane = tid & 63, w = tid >> 6, li = lane & 15, lk = lane >> 4;      // C layout of the MFMA update (front_step)
  const bool has[2] = {t.cld[0] > 0, t.cld[1] > 0};
  const int cld[2] = {t.cld[0], t.cld[1]};
  const int* __restrict__ inv0 = pinv + (has[0] ? t.iofs : 0);
  const int* __restrict__ inv1 = pinv + (has[1] ? t.iofs + ld : 0);
  const double* __restrict__ B0 = fronts_ro + t.boff[0];
  const double* __restrict__ B1 = fronts_ro + t.boff[1];
  // The loads of this kernel form dependent chains (descriptor -> index lists -> data); everything that only needs the
  // descriptor is requested NOW -- the first batch of assembly indices and the right-hand-side permutation next to the index
  // maps, their values next to the children's entries -- so the chain is three round trips deep instead of seven.
  int apos[4], asrc[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int k = t.a0 + u * TB + tid;
    apos[u] = (k < t.a1) ? asm_pos[k] : -1;
    asrc[u] = (k < t.a1) ? asm_src[k] : 0;
  }
  const bool rhs_tile = (nf >= r0 && nf < r0 + TS);      // this tile holds the right-hand-side row
  const int prm = (rhs_tile && tid < ns) ? perm[t.first + tid] : -1;
  // index maps of the tile's rows, columns and of the pivot columns (parent front index -> child boundary index)
  for (int idx = tid; idx < 2 * (2 * TS + PB); idx += TB) {
    const int s = idx / (2 * TS + PB), q = idx % (2 * TS + PB);
    const int* __restrict__ iv = s ? inv1 : inv0;
    if (q < TS) rowI[s][q] = (has[s] && r0 + q <= nf) ? iv[r0 + q] : -1;
    else if (q < 2 * TS) rowJ[s][q - TS] = (has[s] && c0 + q - TS <= nf) ? iv[c0 + q - TS] : -1;
    else piv[s][q - 2 * TS] = (has[s] && q - 2 * TS < ns) ? iv[q - 2 * TS] : -1;
  }
  double aval[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) aval[u] = vals[asrc[u]];
  const double bval = (prm >= 0) ? b[prm] : 0.0;
  __syncthreads();
  // gather (every load issued before it is consumed, clamped to a valid address when the entry has no contribution)
  auto child = [&](int a0, int q0, int a1, int q1) {
    const bool ok0 = a0 >= 0 && q0 >= 0, ok1 = a1 >= 0 && q1 >= 0;
    const double x0 = B0[ok0 ? (long long)cld[0] * q0 + a0 : 0];
    const double x1 = B1[ok1 ? (long long)cld[1] * q1 + a1 : 0];
    return (ok0 ? x0 : 0.0) + (ok1 ? x1 : 0.0);
  };
  double c[4][4];
#pragma unroll
  for (int bj = 0; bj < 4; ++bj)
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const int r = 16 * w + li, cc = 16 * bj + lk + 4 * reg, i = r0 + r, j = c0 + cc;
      const bool in = (i <= nf && j < nf && i >= j);
      c[bj][reg] = in ? child(rowI[0][r], rowJ[0][cc], rowI[1][r], rowJ[1][cc]) : 0.0;
    }
  {
    const int r = tid % TS, qg = tid / TS;      // 64 rows x 4 groups of 8 pivot columns
    double vi[8], vj[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int q = qg * 8 + u;
      vi[u] = vj[u] = 0.0;
      if (!NARROW || qg == 0) {      // wave-uniform
        vi[u] = child(rowI[0][r], piv[0][q], rowI[1][r], piv[1][q]);
        if (!diag) vj[u] = child(rowJ[0][r], piv[0][q], rowJ[1][r], piv[1][q]);
      }
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      ATI[(qg * 8 + u) * TP + r] = vi[u];
      if (!diag) ATJ[(qg * 8 + u) * TP + r] = vj[u];
    }
    const int i = tid % PB, jg = tid / PB;      // pivot block: 32 rows x 8 groups of 4 columns
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int j = jg * 4 + u;
      D[i * LP + j] = (j <= i) ? child(piv[0][i], piv[0][j], piv[1][i], piv[1][j]) : 0.0;
    }
  }
  __syncthreads();
  STAMP(1);
  // assembled entries (all in pivot columns) and right-hand side of the pieces this tile holds; the first batch is in
  // registers already
  auto add_entry = [&](int pos, double v) {
    if (pos < 0) return;
    const int col = pos / ld, row = pos - col * ld;
    if (row < ns) D[row * LP + col] += v;
    else {
      if (row >= r0 && row < r0 + TS) ATI[col * TP + row - r0] += v;
      if (!diag && row >= c0 && row < c0 + TS) ATJ[col * TP + row - c0] += v;
    }
  };
#pragma unroll
  for (int u = 0; u < 4; ++u) add_entry(apos[u], aval[u]);
  for (int k0a = t.a0 + 4 * TB; k0a < t.a1; k0a += 4 * TB) {      // further batches (large fronts only)
    int pos[4], src[4];
    double v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int k = k0a + u * TB + tid;
      pos[u] = (k < t.a1) ? asm_pos[k] : -1;
      src[u] = (k < t.a1) ? asm_src[k] : 0;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = vals[src[u]];
#pragma unroll
    for (int u = 0; u < 4; ++u) add_entry(pos[u], v[u]);
  }
  if (prm >= 0) ATI[tid * TP + nf - r0] += bval;      // ns <= 32 <= TB: one entry per thread
  __syncthreads();
  factor_diag_block<NARROW, DENSE>(D, kw, Lo, (t.ti == 0 && t.tj == 0) ? linv + t.loff : nullptr, fail, nullptr);
  if (!DENSE) {
    for (int idx = tid; idx < PB * PB; idx += TB) Lc[lq_index(idx % PB, idx / PB)] = Lo[(idx % PB) * LP + idx / PB];     // L[m][j], quad order
    __syncthreads();
  }
  STAMP(2);
  const double* Lf = DENSE ? Lo : Lc;
  if (diag) trsm_block64<NARROW, DENSE>(ATI, TP, Lf, kw);      // all four waves, four lanes per row (see front_step)
  else trsm_block64_pair<NARROW, DENSE>(ATI, ATJ, TP, Lf, kw);
  __syncthreads();
  if (t.tj == min((int)t.ti, max(1, (nf - k1 + TS - 1) / TS) - 1)) {      // the row block's last (diagonal) tile, see front_step
    for (int idx = tid; idx < TS * PB; idx += TB) {
      const int r = idx / PB, m = idx % PB;
      if (r0 + r <= nf && m < kw) F[(long long)ld * (r0 + r) + m] = ATI[m * TP + r];
    }
  }
  STAMP(3);
  // rank-32 (rank-8 when NARROW) update on the matrix cores, operand and result layout as in front_step
  typedef double v4f64 __attribute__((ext_vector_type(4)));
  const double* LI = ATI + 16 * w + li + lk * TP;
  const double* LJ = (diag ? ATI : ATJ) + li + lk * TP;
  v4f64 acc[4];      // starts as C; the negated L_I operand makes the matrix core return C - L_I L_J'
#pragma unroll
  for (int bj = 0; bj < 4; ++bj) acc[bj] = v4f64{c[bj][0], c[bj][1], c[bj][2], c[bj][3]};
#pragma unroll
  for (int ks = 0; ks < (NARROW ? 2 : PB / 4); ++ks) {
    const double bv = -LI[4 * ks * TP];
#pragma unroll
    for (int bj = 0; bj < 4; ++bj)
      acc[bj] = __builtin_amdgcn_mfma_f64_16x16x4f64(LJ[4 * ks * TP + 16 * bj], bv, acc[bj], 0, 0, 0);
  }
#pragma unroll
  for (int bj = 0; bj < 4; ++bj)
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const int i = r0 + 16 * w + li, j = c0 + 16 * bj + lk + 4 * reg;
      if (i <= nf && j < nf && i >= j) F[(long long)ld * j + i] = acc[bj][reg];
    }
  STAMP(7);
}

template <bool NARROW>
__global__ __launch_bounds__(TB) __attribute__((amdgpu_waves_per_eu(2, 2))) void front_single_kernel(
    const SingleTile* __restrict__ tiles, const int* __restrict__ pinv, const int* __restrict__ asm_src,
    const int* __restrict__ asm_pos, const double* __restrict__ vals, const int* __restrict__ perm, const double* __restrict__ b,
    const double* __restrict__ fronts_ro, double* fronts, double* linv, int* fail, long long* prof) {
  front_single_body<NARROW, false>(tiles, pinv, asm_src, asm_pos, vals, perm, b, fronts_ro, fronts, linv, fail, prof);
}

template <bool NARROW>
__global__ __launch_bounds__(TB) __attribute__((amdgpu_waves_per_eu(3, 3))) void front_single_dense_kernel(
    const SingleTile* __restrict__ tiles, const int* __restrict__ pinv, const int* __restrict__ asm_src,
    const int* __restrict__ asm_pos, const double* __restrict__ vals, const int* __restrict__ perm, const double* __restrict__ b,
    const double* __restrict__ fronts_ro, double* fronts, double* linv, int* fail, long long* prof) {
  front_single_body<NARROW, true>(tiles, pinv, asm_src, asm_pos, vals, perm, b, fronts_ro, fronts, linv, fail, prof);
}

// Leaf heights (no children) whose fronts fit LDS and whose trailing matrix after the first panel is one 64x64
// tile (nf <= 95, ns <= 64; at fem2d L=7 the 1 024 leaves, nf ~ 64, ns ~ 35): ONE workgroup assembles the front in
// LDS, runs its (at most two) panels -- pivot-block factor, panel rows by in-register substitution, rank-32
// update on the 4x4 micro-tiles of the tile kernels -- and writes the Schur complement, the rows of L and the
// pivot blocks back.  One launch replaces front_start + two front_step launches of the widest height of the tree.
__global__ __launch_bounds__(TB, 4) void front_leaf_kernel(
    const GNode* __restrict__ nodes, const int* __restrict__ list, const int* __restrict__ asm_src,
    const int* __restrict__ asm_pos, const double* __restrict__ vals, const int* __restrict__ perm,
    const double* __restrict__ b, double* fronts, double* linv, int* fail, long long* prof) {
  extern __shared__ __attribute__((aligned(32))) double sm[];
  STAMP(0);
  const GNode nd = nodes[blockIdx.x];      // per-launch copy in launch order: one descriptor round trip, not two
  (void)list;
  const int nf = nd.nf, ld = nf + 1, ns = nd.ns, tid = threadIdx.x;
  constexpr int TP = TS + 8;      // row stride of the staged panel rows (as in front_step)
  __shared__ __attribute__((aligned(32))) double fixed[TP * PB + PB * (PB + 1) / 2];      // AT (also D) | Lo (packed)
  double* Fs = sm;                               // the front, (nf+1)-leading-dimension layout as in HBM
  double* AT = fixed;                            // staging tile of the panel rows (front_step layout), 32 x 64
  double* D = fixed;                             // the pivot block is dead once factored: same storage
  double* Lo = fixed + TP * PB;                  // row-major packed factor, reciprocal diagonal
  // packed lower-triangular storage of the front (column j holds rows j..nf): half the LDS of the square layout,
  // which (with the packed pivot-block factor) is what lets four workgroups share a CU
  auto P = [ld](int i, int j) { return j * ld - (j * (j - 1)) / 2 + (i - j); };
  const int total = nf * ld - (nf * (nf - 1)) / 2;
  double* F = fronts + nd.off;
  for (int idx = tid; idx < total; idx += TB) Fs[idx] = 0.0;
  __syncthreads();
  for (int k = nd.a0 + tid; k < nd.a1; k += TB) {
    const int pos = asm_pos[k], col = pos / ld;
    Fs[P(pos - col * ld, col)] += vals[asm_src[k]];
  }
  for (int c = tid; c < ns; c += TB) Fs[P(nf, c)] += b[perm[nd.first + c]];
  __syncthreads();
  STAMP(1);
  auto panel = [&](const int p) {      // straight-line code for the (at most two) panels: a loop here made the
                                        // compiler spill the substitution's registers
    const int k0 = p * PB, kw = min(PB, ns - k0), k1 = k0 + kw;
    for (int idx = tid; idx < PB * PB; idx += TB) {
      const int i = idx % PB, j = idx / PB;
      D[i * LP + j] = (i < kw && j <= i) ? Fs[P(k0 + i, k0 + j)] : 0.0;
    }
    __syncthreads();
    factor_diag_block<true, true>(D, kw, Lo, linv + nd.loff + (long long)p * 2 * PB * PB, fail, nullptr);
    // the (at most 64) panel rows k1..nf go through the transposed staging tile of the tile kernels, so that the
    // substitution and the rank-32 update are literally the code of front_step (constant LDS strides)
    for (int idx = tid; idx < TS * PB; idx += TB) {
      const int r = idx % TS, q = idx / TS;
      AT[q * TP + r] = (q < kw && k1 + r <= nf) ? Fs[P(k1 + r, k0 + q)] : 0.0;
    }
    __syncthreads();
    trsm_block64<true, true>(AT, TP, Lo, kw);      // all four waves, four lanes per row, straight from the packed factor
    __syncthreads();
    for (int idx = tid; idx < TS * PB; idx += TB) {      // mirrored L for the backward sweep: row k1 + r, kw consecutive entries
      const int r = idx / PB, m = idx % PB;
      if (k1 + r <= nf && m < kw) F[(long long)ld * (k1 + r) + k0 + m] = AT[m * TP + r];
    }
    // rank-kw update of the 64x64 trailing tile on the matrix cores (operand / result layout of front_step)
    typedef double v4f64 __attribute__((ext_vector_type(4)));
    const int lane = tid & 63, w = tid >> 6, li = lane & 15, lk = lane >> 4;
    const double* LI = AT + 16 * w + li + lk * TP;
    const double* LJ = AT + li + lk * TP;
    v4f64 acc[4];
#pragma unroll
    for (int bj = 0; bj < 4; ++bj) acc[bj] = v4f64{0.0, 0.0, 0.0, 0.0};
#pragma unroll 1
    for (int ks = 0; ks < (kw + 3) / 4; ++ks) {      // staged rows >= kw are zero
      const double bv = LI[4 * ks * TP];
#pragma unroll
      for (int bj = 0; bj < 4; ++bj)
        acc[bj] = __builtin_amdgcn_mfma_f64_16x16x4f64(LJ[4 * ks * TP + 16 * bj], bv, acc[bj], 0, 0, 0);
    }
#pragma unroll
    for (int bj = 0; bj < 4; ++bj)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int i = k1 + 16 * w + li, j = k1 + 16 * bj + lk + 4 * reg;
        if (i <= nf && j < nf && i >= j) Fs[P(i, j)] -= acc[bj][reg];
      }
    __syncthreads();
  };
  panel(0);
  if (ns > PB) panel(1);
  STAMP(2);
  // Schur complement + reduced right-hand side for the parent: boundary columns, rows down to nf
  const int nb = nf - ns;
  for (int idx = tid; idx < (nb + 1) * nb; idx += TB) {
    const int i = ns + idx % (nb + 1), j = ns + idx / (nb + 1);
    if (i >= j) F[(long long)ld * j + i] = Fs[P(i, j)];
  }
  STAMP(7);
}

// s_j = sum_i L[ns+i][j] x_bdry[i] for the own columns j of one front: thread = (column, slice of the boundary
// rows); L[i][j] sits at row j of column i, so the loads of a wave are unit-stride and independent.
template <int NT>
__device__ inline void rect_part(const double* __restrict__ F, int ld, int ns, int nb, int j0, int jw,
                                 const double* xb, double* red, double* out /* LDS or global, indexed by j */, bool subtract) {
  const int tid = threadIdx.x;
  const int nsl = NT / jw, jl = tid % jw, sl = tid / jw, j = j0 + jl;
  double s = 0.0;
  if (sl < nsl && j < ns) {
    const double* col = F + (long long)ld * ns + j;
#pragma unroll 8
    for (int i = sl; i < nb; i += nsl) s = fma(col[(long long)ld * i], xb[i], s);
  }
  if (nsl > 1) {
    red[tid] = s;
    __syncthreads();
    if (sl == 0)
      for (int q = 1; q < nsl; ++q) s += red[q * jw + jl];
  }
  if (sl == 0 && j < ns) out[j] = subtract ? out[j] - s : s;
}

__global__ __launch_bounds__(RT) void backward_rect_kernel(const GNode* __restrict__ nodes, const RectJob* __restrict__ jobs,
                                                            const int* __restrict__ bdry_all,
                                                            const double* __restrict__ fronts, const double* __restrict__ y,
                                                            double* rect) {
  extern __shared__ double sh[];      // xb[nb] | red[RT]
  const RectJob job = jobs[blockIdx.x];
  const GNode nd = nodes[blockIdx.x];      // per-job copy of the node (no second descriptor round trip)
  const int nf = nd.nf, ns = nd.ns, nb = nf - ns;
  const int* bd = bdry_all + nd.bofs;
  double* xb = sh;
  double* red = sh + nb;
  for (int i = threadIdx.x; i < nb; i += RT) xb[i] = y[bd[i]];
  __syncthreads();
  rect_part<RT>(fronts + nd.off, nf + 1, ns, nb, job.chunk * 64, 64, xb, red, rect + nd.first, false);
}

// Backward sweep L' x = u of one height (heights descending, ancestors' entries of x are final): the forward
// result u sits in row nf of the factored front (= column nf of the mirrored L).  Right-looking over panels:
// x_p = L_pp^-T u_p by in-wave substitution, then u_j -= sum_{i in p} L[i][j] x_i for all earlier j
// (thread per j, unit-stride loads, no reduction unless the panel rows are split over idle threads).
template <int NT>
__global__ __launch_bounds__(NT) void backward_kernel(const GNode* __restrict__ nodes, const int* __restrict__ list,
                                                       const int* __restrict__ bdry_all, const double* __restrict__ fronts,
                                                       const double* __restrict__ linv, const int* __restrict__ perm,
                                                       const double* __restrict__ rect, int use_rect, double* y, double* x,
                                                       long long* prof) {
  extern __shared__ double sh[];      // u[ns] | xb[nb] | red[NT]
  STAMP(0);      // stamps: 1 right-hand side staged, 2 boundary part subtracted, 3 / 4 / 5 the last panel's substitution, its
                 // barrier, its update of the earlier columns, 6 all panels done, 7 solution stored
  const GNode nd = nodes[blockIdx.x];      // per-launch copy in launch order: one descriptor round trip, not two
  (void)list;
  const int nf = nd.nf, ns = nd.ns, nb = nf - ns, ld = nf + 1, tid = threadIdx.x;
  const double* F = fronts + nd.off;
  double* u = sh;
  double* xb = sh + ns;
  double* red = sh + nf;
  for (int i = tid; i < ns; i += NT) u[i] = F[(long long)ld * nf + i] - ((use_rect && nb > 0) ? rect[nd.first + i] : 0.0);
  STAMP(1);
  if (!use_rect && nb > 0) {      // workgroup-uniform
    const int* bd = bdry_all + nd.bofs;
    for (int i = tid; i < nb; i += NT) xb[i] = y[bd[i]];
    __syncthreads();
    const int jw = min(NT, (ns + 63) & ~63);
    for (int j0 = 0; j0 < ns; j0 += jw) {
      rect_part<NT>(F, ld, ns, nb, j0, jw, xb, red, u, true);
      __syncthreads();
    }
  }
  __syncthreads();
  STAMP(2);
  const int npanel = (ns + PB - 1) / PB;
  for (int pp = npanel - 1; pp >= 0; --pp) {
    const int k0 = pp * PB, kw = min(PB, ns - k0), k1 = k0 + kw;
    if (tid < 64) {
      // x_p = L_pp^-T u_p by wave 0: lane c owns x_c and column c of L_pp (row m of the stored block is
      // unit-stride over c); right-looking from the last row, x_m broadcast with constant-lane v_readlane
      const int c = tid & 31;
      const double* lpp = linv + nd.loff + (long long)pp * 2 * PB * PB + c;
      double lc[PB];
#pragma unroll
      for (int m = 0; m < PB; ++m) lc[m] = lpp[PB * m];          // L[m][c] (1/L[c][c] on the diagonal, 0 above)
      double uc = (c < kw) ? u[k0 + c] : 0.0;
#pragma unroll
      for (int m = PB - 1; m >= 0; --m) {
        const double xm = readlane_f64(uc * lc[m], m);            // lane m: u_m / L[m][m]
        uc = (c == m) ? xm : fma(-lc[m], xm, uc);                 // lanes c > m: lc[m] == 0
      }
      if (tid < kw) u[k0 + c] = uc;
      if (pp == npanel - 1) STAMP(3);
    }
    __syncthreads();
    if (pp == npanel - 1) STAMP(4);
    if (k0 > 0) {
      const int jw = min(NT, (k0 + 63) & ~63), nsl = NT / jw;
      for (int jb = 0; jb < k0; jb += jw) {
        const int jl = tid % jw, sl = tid / jw, j = jb + jl;
        double s = 0.0;
        if (sl < nsl && j < k0) {
#pragma unroll 8
          for (int i = k0 + sl; i < k1; i += nsl) s = fma(F[(long long)ld * i + j], u[i], s);
        }
        if (nsl > 1) {      // then jw >= k0: a single jb iteration
          red[tid] = s;
          __syncthreads();
          if (sl == 0)
            for (int q = 1; q < nsl; ++q) s += red[q * jw + jl];
        }
        if (sl == 0 && j < k0) u[j] -= s;
      }
      __syncthreads();
    }
    if (pp == npanel - 1) STAMP(5);
  }
  STAMP(6);
  for (int i = tid; i < ns; i += NT) {
    const double v = u[i];
    y[nd.first + i] = v;
    x[perm[nd.first + i]] = v;
  }
  STAMP(7);
}

// Split factorisation: the Schur complement of a subtree root -- lower triangle of the boundary block of its front plus
// the right-hand-side row, column j packed at j (nb + 1) - j (j - 1) / 2 -- into / out of the exchange buffer.  A rank
// that does not own the subtree contributes zeros, so the sum over the ranks is the owner's block, exactly.
__global__ __launch_bounds__(256) void schur_pack_kernel(const RootXchg* __restrict__ roots, const double* __restrict__ fronts,
                                                         double* __restrict__ xb) {
  const RootXchg r = roots[blockIdx.x];
  const double* F = fronts + r.off;
  for (int j = blockIdx.y; j < r.nb; j += gridDim.y) {
    double* dst = xb + r.xoff + (long long)j * (r.nb + 1) - (long long)j * (j - 1) / 2 - j;      // dst[i], i = j .. nb
    const double* col = F + (long long)r.ld * (r.ns + j) + r.ns;
    for (int i = j + threadIdx.x; i <= r.nb; i += blockDim.x) dst[i] = r.owned ? col[i] : 0.0;
  }
}

__global__ __launch_bounds__(256) void schur_unpack_kernel(const RootXchg* __restrict__ roots, const double* __restrict__ xb,
                                                           double* __restrict__ fronts) {
  const RootXchg r = roots[blockIdx.x];
  double* F = fronts + r.off;
  for (int j = blockIdx.y; j < r.nb; j += gridDim.y) {
    const double* src = xb + r.xoff + (long long)j * (r.nb + 1) - (long long)j * (j - 1) / 2 - j;
    double* col = F + (long long)r.ld * (r.ns + j) + r.ns;
    for (int i = j + threadIdx.x; i <= r.nb; i += blockDim.x) col[i] = src[i];
  }
}

// values_local(): the matrix entries of the top nodes, gathered behind the Schur complements and scattered back summed
__global__ __launch_bounds__(256) void vals_pack_kernel(int n, const int* __restrict__ idx, const double* __restrict__ vals,
                                                        double* __restrict__ xb) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) xb[i] = vals[idx[i]];
}

__global__ __launch_bounds__(256) void vals_unpack_kernel(int n, const int* __restrict__ idx, const double* __restrict__ xb,
                                                          double* __restrict__ vals) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) vals[idx[i]] = xb[i];
}

// x of the unknowns this rank is responsible for (its subtree; rank 0 also the top), zeros elsewhere, + the pivot flag
__global__ __launch_bounds__(256) void xsol_pack_kernel(int n, const int* __restrict__ own, const double* __restrict__ x,
                                                        const int* __restrict__ fail, double* __restrict__ xs) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i <= n; i += (long long)gridDim.x * blockDim.x)
    xs[i] = (i == n) ? (*fail ? 1.0 : 0.0) : (own[i] ? x[i] : 0.0);
}

// owner-local solution: keep x on this rank's own unknowns and on the top, zero elsewhere
__global__ __launch_bounds__(256) void x_local_kernel(int n, const int* __restrict__ kind, double* __restrict__ x) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    if (kind[i] == 0) x[i] = 0.0;
}

__global__ __launch_bounds__(256) void xsol_unpack_kernel(int n, const double* __restrict__ xs, double* __restrict__ x, int* fail) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i <= n; i += (long long)gridDim.x * blockDim.x) {
    if (i == n) *fail = xs[n] != 0.0 ? 1 : 0;
    else x[i] = xs[i];
  }
}

}  // namespace

template <class T>
T* GpuChol::upload(const std::vector<T>& v) {
  T* d = nullptr;
  const size_t bytes = std::max<size_t>(v.size(), 1) * sizeof(T);
  ck(hipMalloc((void**)&d, bytes), "hipMalloc");
  allocs_.push_back(d);
  if (!v.empty()) ck(hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice), "H2D");
  return d;
}

GpuChol::~GpuChol() {
  for (GraphEntry& g : graphs_) (void)hipGraphExecDestroy(g.exec);
  for (void* p : allocs_) (void)hipFree(p);
}

void GpuChol::build(const MfChol& sym, Ctx* ctx) {
  ctx_ = ctx;
  part_ = (ctx && ctx->world > 1) ? sym.partition(ctx->world) : CholPartition();
  if (!part_.split()) part_.owner.assign(sym.nodes_.size(), -1);
  const int my_rank = ctx ? ctx->rank : 0;
  n_ = sym.n_;
  nnodes_ = (int)sym.nodes_.size();
  flops_ = sym.flops_;
  std::vector<GNode> nodes(nnodes_);
  std::vector<int> bdry_all, ea_all, pinv, height(nnodes_, 0);
  long long off = 0, loff = 0;
  max_nf_ = 0;
  for (int t = 0; t < nnodes_; ++t) {
    const auto& nd = sym.nodes_[t];
    GNode& g = nodes[t];
    g.off = off;
    g.loff = loff;
    loff += (long long)((nd.ns + PB - 1) / PB) * 2 * PB * PB;
    g.nf = nd.nf();
    g.ns = nd.ns;
    g.first = nd.first;
    g.parent = nd.parent;
    g.bofs = (int)bdry_all.size();
    g.child[0] = nd.children.size() > 0 ? nd.children[0] : -1;
    g.child[1] = nd.children.size() > 1 ? nd.children[1] : -1;
    g.iofs = -1;
    if (nd.children.size() > 2) throw InternalError("gpuchol: elimination tree is not binary");
    bdry_all.insert(bdry_all.end(), nd.bdry.begin(), nd.bdry.end());
    ea_all.insert(ea_all.end(), nd.ea.begin(), nd.ea.end());
    if (nd.parent >= 0 && nd.ea.size() != nd.bdry.size()) throw InternalError("gpuchol: ea/bdry size mismatch");
    if (nd.parent >= 0 && !std::is_sorted(nd.ea.begin(), nd.ea.end())) throw InternalError("gpuchol: ea not ascending");
    if (nd.parent < 0) ea_all.resize(bdry_all.size(), 0);
    if ((long long)(g.nf + 1) * (g.nf + 1) > 2000000000LL) throw ArgError("gpuchol: front too large");
    off += (long long)(g.nf + 1) * (g.nf + 1);
    max_nf_ = std::max(max_nf_, g.nf);
    for (int c : nd.children) height[t] = std::max(height[t], height[c] + 1);   // postorder: children first
  }
  total_front_ = off;
  // per parent and child slot: parent front row -> child boundary row (the child's right-hand-side row for nf)
  for (int t = 0; t < nnodes_; ++t) {
    GNode& g = nodes[t];
    if (g.child[0] < 0 && g.child[1] < 0) continue;
    g.iofs = (int)pinv.size();
    pinv.resize(pinv.size() + 2 * (size_t)(g.nf + 1), -1);
    for (int s = 0; s < 2; ++s) {
      const int c = g.child[s];
      if (c < 0) continue;
      int* iv = pinv.data() + g.iofs + (size_t)s * (g.nf + 1);
      const int cnb = nodes[c].nf - nodes[c].ns;
      const int* ea = ea_all.data() + nodes[c].bofs;
      for (int a = 0; a < cnb; ++a) {
        if (ea[a] < 0 || ea[a] >= g.nf || iv[ea[a]] != -1) throw InternalError("gpuchol: bad extend-add map");
        iv[ea[a]] = a;
      }
      iv[g.nf] = cnb;
    }
  }
  nheights_ = nnodes_ ? *std::max_element(height.begin(), height.end()) + 1 : 0;
  // assembly map, per node sorted by the front_start job (32-column chunk, 256-row block counted from the chunk's
  // first row) that owns the destination; positions in the (nf+1)-leading-dimension layout
  std::vector<int> asrc, apos;
  std::vector<std::vector<StartJob>> sjobs(nnodes_);
  for (int t = 0; t < nnodes_; ++t) {
    const int nf = nodes[t].nf, ld = nf + 1;
    const size_t m = sym.a_idx_[t].size();
    const int nch = (nf + PB - 1) / PB;
    auto key = [&](int e) {
      const int pos = sym.a_pos_[t][e], col = pos / nf, row = pos % nf, ch = col / PB;
      return (long long)ch * 65536 + (row - ch * PB) / TB;
    };
    std::vector<int> ord(m);
    std::iota(ord.begin(), ord.end(), 0);
    std::stable_sort(ord.begin(), ord.end(), [&](int a, int b) { return key(a) < key(b); });
    size_t q = 0;
    for (int ch = 0; ch < nch; ++ch) {
      const int nrb = (nf + 1 - ch * PB + TB - 1) / TB;      // rows ch*32 .. nf inclusive
      if (nrb > 65535) throw ArgError("gpuchol: front too large for the start-job key");
      for (int rb = 0; rb < nrb; ++rb) {
        StartJob j{};
        j.node = t;
        j.chunk = ch;
        j.rb = rb;
        j.off = nodes[t].off;
        j.loff = nodes[t].loff;
        j.nf = nodes[t].nf;
        j.ns = nodes[t].ns;
        j.iofs = nodes[t].iofs;
        j.first = nodes[t].first;
        for (int sI = 0; sI < 2; ++sI) {
          j.boff[sI] = nodes[t].off;
          j.cld[sI] = 0;
          if (nodes[t].child[sI] >= 0) {
            const GNode& c = nodes[nodes[t].child[sI]];
            j.cld[sI] = c.nf + 1;
            j.boff[sI] = c.off + (long long)j.cld[sI] * c.ns + c.ns;
          }
        }
        j.a0 = (int)asrc.size();
        while (q < m && key(ord[q]) == (long long)ch * 65536 + rb) {
          const int pos = sym.a_pos_[t][ord[q]], col = pos / nf, row = pos % nf;
          if (row < col) throw InternalError("gpuchol: assembly entry above the diagonal");
          asrc.push_back(sym.a_idx_[t][ord[q]]);
          apos.push_back(ld * col + row);
          ++q;
        }
        j.a1 = (int)asrc.size();
        sjobs[t].push_back(j);
      }
    }
    if (q != m) throw InternalError("gpuchol: assembly entry outside the front");
    nodes[t].a0 = sjobs[t].empty() ? 0 : sjobs[t].front().a0;
    nodes[t].a1 = sjobs[t].empty() ? 0 : sjobs[t].back().a1;
  }
  // schedule
  static const bool leaf_ok = [] {
    const char* e = std::getenv("MGB_CHOL_LEAF");
    return !(e && e[0] == '0');
  }();
  static const bool single_ok = [] {
    const char* e = std::getenv("MGB_CHOL_SINGLE");
    return !(e && e[0] == '0');
  }();
  static const bool step2_ok = [] {      // MGB_CHOL_STEP2=0: one panel per launch everywhere (the scheme before front_step2)
    const char* e = std::getenv("MGB_CHOL_STEP2");
    return !(e && e[0] == '0');
  }();
  static const bool start_pivot_ok = [] {      // MGB_CHOL_START_PIVOT=0: the first pivot block is factored by the assembling workgroup
    const char* e = std::getenv("MGB_CHOL_START_PIVOT");
    return !(e && e[0] == '0');
  }();
  start_pivot_ = start_pivot_ok;
  static const bool wide_ok = [] {      // MGB_CHOL_WIDE=0: one panel per launch beyond the fused kernel's tile count (the scheme before)
    const char* e = std::getenv("MGB_CHOL_WIDE");
    return !(e && e[0] == '0');
  }();
  static const int step2_max_tiles = [] {
    const char* e = std::getenv("MGB_CHOL_STEP2_TILES");
    return e ? std::atoi(e) : 224;
  }();
  static const int split_nf = [] {
    const char* e = std::getenv("MGB_BWD_SPLIT_NF");
    return e ? std::atoi(e) : 192;
  }();
  std::vector<SingleTile> singles;
  std::vector<int> lists;
  std::vector<StartJob> starts;
  std::vector<StepTile> tiles;
  std::vector<RectJob> rects;
  launches_ = 0;
  // one schedule per node set: `which` = 0 own nodes (split: this rank's subtree; else everything), 1 = the replicated top
  auto make_plans = [&](int which, std::vector<HeightPlan>& out) {
  out.clear();
  for (int h = 0; h < nheights_; ++h) {
    HeightPlan hp_new;
    HeightPlan& hp = hp_new;
    hp.nodes.ofs = (int)lists.size();
    int max_ns = 0;
    hp.max_nf = 0;
    std::vector<int> mine;
    for (int t = 0; t < nnodes_; ++t)
      if (height[t] == h && (part_.split() ? (which == 0 ? part_.owner[t] == my_rank : part_.owner[t] < 0) : which == 0)) {
        mine.push_back(t);
        max_ns = std::max(max_ns, nodes[t].ns);
        hp.max_nf = std::max(hp.max_nf, nodes[t].nf);
      }
    if (mine.empty()) continue;
    lists.insert(lists.end(), mine.begin(), mine.end());
    hp.nodes.cnt = (int)mine.size();
    // leaf heights with small fronts: the whole front in one workgroup (front_leaf_kernel)
    {
      bool leaf = leaf_ok && hp.max_nf <= 95 && max_ns <= 2 * PB;
      for (int t : mine) leaf = leaf && nodes[t].child[0] < 0 && nodes[t].child[1] < 0 && nodes[t].ns >= 1;
      hp.leaf = leaf && !mine.empty();
    }
    // single-panel heights with children: one dependency-free launch (front_single_kernel)
    bool any_child = false, all_pivots = true;
    for (int t : mine) {
      any_child = any_child || nodes[t].child[0] >= 0 || nodes[t].child[1] >= 0;
      all_pivots = all_pivots && nodes[t].ns >= 1;      // a pass-through front (ns = 0) needs front_start to copy it
    }
    hp.single = single_ok && max_ns <= PB && any_child && all_pivots;
    hp.narrow = hp.single && max_ns <= 8;
    // front_start jobs
    hp.start.ofs = (int)starts.size();
    hp.start_bytes = 0;
    for (int t : mine) {
      const GNode& g = nodes[t];
      hp.start_bytes += 0.5 * g.nf * g.nf * 8.0 + (double)sym.a_idx_[t].size() * 20.0;
      for (int s = 0; s < 2; ++s) {
        const int c = g.child[s];
        if (c < 0) continue;
        const double cnb = nodes[c].nf - nodes[c].ns;
        hp.start_bytes += 0.5 * cnb * cnb * 8.0;     // child entry read (the parent entry write is counted above)
      }
    }
    if (!hp.single && !hp.leaf) {
      if (start_pivot_ok)      // dedicated pivot jobs first: they are what the first panel launch waits for
        for (int t : mine)
          if (nodes[t].ns > 0 && !sjobs[t].empty()) {
            StartJob pj = sjobs[t].front();      // chunk 0, row block 0: its assembly range covers the pivot block
            pj.rb = -1;
            starts.push_back(pj);
          }
      for (int t : mine) starts.insert(starts.end(), sjobs[t].begin(), sjobs[t].end());
    }
    hp.start.cnt = (int)starts.size() - hp.start.ofs;
    launches_++;
    // front_step tiles, pivot-owning (0,0) tiles first
    const int npanel = hp.leaf ? 0 : (max_ns + PB - 1) / PB;
    for (int p = 0; p < npanel; ++p) {
      Range rt{(int)tiles.size(), 0};
      double bytes = 0;
      int npiv = 0;
      // two panels per launch (front_step2) while the height has at least two left; the odd last one runs front_step
      // ... and only where the launch is latency-bound (its tiles fit the chip in one round): front_step2 holds 115 KB of
      // LDS, one workgroup per CU, and loses against two rank-32 launches at 2-3 workgroups per CU on the big 3-D heights
      int ntile2 = 0;
      for (int t : mine) {
        const GNode& g = nodes[t];
        if (g.ns <= p * PB) continue;
        const int k2 = std::min(g.ns, (p + 2) * PB), Tr = (g.nf + 1 - k2 + TS - 1) / TS, Tc = std::max(1, (g.nf - k2 + TS - 1) / TS);
        for (int ti = 0; ti < Tr; ++ti) ntile2 += std::min(ti, Tc - 1) + 1;
      }
      const bool pair = step2_ok && !hp.single && p + 1 < npanel && ntile2 <= step2_max_tiles;
      // beyond that tile count: the same two panels as a panel launch (one workgroup per 64-row block, row block 0 also
      // factors the next pivot block) and an update launch (the tiles, rank-64, nothing else) -- see front_step2_body
      const bool wide = wide_ok && step2_ok && !hp.single && p + 1 < npanel && !pair;
      if (wide) {
        double pbytes = 0, ubytes = 0;
        for (int t : mine) {
          const GNode& g = nodes[t];
          if (g.ns <= p * PB) continue;
          const int k2 = std::min(g.ns, (p + 2) * PB), Tr = (g.nf + 1 - k2 + TS - 1) / TS;
          StepTile st{};
          st.off = g.off;
          st.loff = g.loff + (long long)p * 2 * PB * PB;
          st.nf = g.nf;
          st.ns = g.ns;
          st.pad = t;
          (void)Tr;
          tiles.push_back(st);      // the front's pivot workgroup: trailing rows k2 .. k2 + 31
          npiv++;
          const double tr = g.nf + 1 - k2;
          pbytes += tr * (k2 - p * PB) * 32.0;      // panel read, mirrored + in-place write
        }
        for (int t : mine) {
          const GNode& g = nodes[t];
          if (g.ns <= p * PB) continue;
          const int k2 = std::min(g.ns, (p + 2) * PB), Trw = std::max(0, (g.nf + 1 - k2 - PB + TS - 1) / TS);
          StepTile st{};
          st.off = g.off;
          st.loff = g.loff + (long long)p * 2 * PB * PB;
          st.nf = g.nf;
          st.ns = g.ns;
          st.pad = t;
          for (int ti = 0; ti < Trw; ++ti) {
            st.ti = st.tj = (short)ti;
            tiles.push_back(st);
          }
        }
        rt.cnt = (int)tiles.size() - rt.ofs;
        hp.step_npiv.push_back(npiv);
        hp.step.push_back(rt);
        hp.step_bytes.push_back(pbytes);
        hp.step_p.push_back(p);
        hp.step_pair.push_back(2);
        launches_++;
        Range ru{(int)tiles.size(), 0};
        for (int t : mine) {
          const GNode& g = nodes[t];
          if (g.ns <= p * PB) continue;
          const int k2 = std::min(g.ns, (p + 2) * PB), Tr = (g.nf + 1 - k2 + TS - 1) / TS, Tc = std::max(1, (g.nf - k2 + TS - 1) / TS);
          if (Tr > 30000) throw ArgError("gpuchol: front too large for tile index");
          StepTile st{};
          st.off = g.off;
          st.loff = g.loff + (long long)p * 2 * PB * PB;
          st.nf = g.nf;
          st.ns = g.ns;
          st.pad = t;
          for (int ti = 0; ti < Tr; ++ti)
            for (int tj = 0; tj <= std::min(ti, Tc - 1); ++tj) {
              st.ti = (short)ti;
              st.tj = (short)tj;
              tiles.push_back(st);
            }
          const double tr = g.nf + 1 - k2;
          ubytes += tr * (k2 - p * PB) * 8.0 + 0.5 * tr * tr * 16.0;      // solved panel read, trailing read + write
        }
        ru.cnt = (int)tiles.size() - ru.ofs;
        hp.step_npiv.push_back(0);
        hp.step.push_back(ru);
        hp.step_bytes.push_back(ubytes);
        hp.step_p.push_back(p);
        hp.step_pair.push_back(3);
        launches_++;
        ++p;
        continue;
      }
      const int np = pair ? 2 : 1;
      for (int pass = 0; pass < 2; ++pass)
        for (int t : mine) {
          const GNode& g = nodes[t];
          if (g.ns <= p * PB) continue;
          const int k1 = std::min(g.ns, (p + np) * PB), kw = k1 - p * PB;      // first trailing row, pivots of this launch
          StepTile st{};
          st.off = g.off;
          st.loff = g.loff + (long long)p * 2 * PB * PB;
          st.nf = g.nf;
          st.ns = g.ns;
          st.pad = t;      // node index (front_single_kernel)
          if (pass == 0) {      // pivot workgroups first: they are the critical path of the next launch
            if (k1 < g.ns) {
              tiles.push_back(st);
              npiv++;
            }
            const double tr = g.nf + 1 - k1;
            bytes += tr * kw * 16.0 + 0.5 * tr * tr * 16.0;      // panel read + mirrored write, trailing read + write
            continue;
          }
          const int Tr = (g.nf + 1 - k1 + TS - 1) / TS, Tc = std::max(1, (g.nf - k1 + TS - 1) / TS);
          if (Tr > 30000) throw ArgError("gpuchol: front too large for tile index");
          for (int ti = 0; ti < Tr; ++ti)
            for (int tj = 0; tj <= std::min(ti, Tc - 1); ++tj) {
              st.ti = (short)ti;
              st.tj = (short)tj;
              tiles.push_back(st);
            }
        }
      hp.step_npiv.push_back(npiv);
      if (hp.single) {      // same tiles, self-contained descriptors
        hp.single_tiles.ofs = (int)singles.size();
        for (int q = rt.ofs; q < (int)tiles.size(); ++q) {
          const GNode& g = nodes[tiles[q].pad];
          SingleTile u{};
          u.off = g.off;
          u.loff = g.loff;
          u.nf = g.nf;
          u.ns = g.ns;
          u.iofs = g.iofs;
          u.a0 = g.a0;
          u.a1 = g.a1;
          u.first = g.first;
          u.ti = tiles[q].ti;
          u.tj = tiles[q].tj;
          for (int sI = 0; sI < 2; ++sI) {
            u.boff[sI] = g.off;
            u.cld[sI] = 0;
            if (g.child[sI] >= 0) {
              const GNode& c = nodes[g.child[sI]];
              u.cld[sI] = c.nf + 1;
              u.boff[sI] = c.off + (long long)u.cld[sI] * c.ns + c.ns;
            }
          }
          singles.push_back(u);
        }
        hp.single_tiles.cnt = (int)singles.size() - hp.single_tiles.ofs;
      }
      rt.cnt = (int)tiles.size() - rt.ofs;
      hp.step.push_back(rt);
      hp.step_bytes.push_back(bytes);
      hp.step_p.push_back(p);
      hp.step_pair.push_back(pair ? 1 : 0);
      launches_++;
      if (pair) ++p;
    }
    // backward
    hp.split = hp.max_nf > split_nf;
    hp.rect.ofs = (int)rects.size();
    hp.rect_bytes = hp.tri_bytes = 0;
    for (int t : mine) {
      const GNode& g = nodes[t];
      const double nb = g.nf - g.ns;
      hp.rect_bytes += nb * g.ns * 8.0 + nb * 12.0;
      hp.tri_bytes += 0.5 * g.ns * g.ns * 8.0 + g.ns * 28.0;
      if (hp.split && nb > 0)
        for (int ch = 0; ch * 64 < g.ns; ++ch) rects.push_back({t, ch});
    }
    hp.rect.cnt = (int)rects.size() - hp.rect.ofs;
    launches_ += 1 + (hp.rect.cnt ? 1 : 0);
    out.push_back(std::move(hp_new));
  }
  };
  make_plans(0, plan_);
  if (part_.split()) make_plans(1, plan_top_);
  if ((size_t)(max_nf_ + RT + PB) * 8 > 150 * 1024) throw ArgError("gpuchol: front exceeds the LDS budget of the sweeps");
  d_nodes_ = upload(nodes);
  d_perm_ = upload(sym.perm_);
  d_bdry_ = upload(bdry_all);
  d_pinv_ = upload(pinv);
  d_asm_src_ = upload(asrc);
  d_asm_pos_ = upload(apos);
  d_lists_ = upload(lists);
  {      // node descriptors in launch order (front_leaf, backward) and per backward_rect job
    std::vector<GNode> hn(lists.size()), rn(rects.size());
    for (size_t k = 0; k < lists.size(); ++k) hn[k] = nodes[lists[k]];
    for (size_t k = 0; k < rects.size(); ++k) rn[k] = nodes[rects[k].node];
    d_hnodes_ = upload(hn);
    d_rnodes_ = upload(rn);
  }
  d_start_ = upload(starts);
  d_tiles_ = upload(tiles);
  d_singles_ = upload(singles);
  d_rectjobs_ = upload(rects);
  if (part_.split()) {
    std::vector<RootXchg> roots;
    long long xoff = 0;
    for (int j = 0; j < part_.world; ++j) {
      const GNode& g = nodes[part_.roots[j]];
      RootXchg r{};
      r.off = g.off;
      r.xoff = xoff;
      r.ld = g.nf + 1;
      r.ns = g.ns;
      r.nb = g.nf - g.ns;
      r.owned = (j == my_rank) ? 1 : 0;
      xoff += (long long)r.nb * (r.nb + 3) / 2;
      max_root_nb_ = std::max(max_root_nb_, r.nb);
      roots.push_back(r);
    }
    xchg_doubles_ = xoff;
    nroots_ = (int)roots.size();
    d_roots_ = upload(roots);
    std::vector<int> own(n_, 0);
    for (int t = 0; t < nnodes_; ++t)
      if (part_.owner[t] == my_rank || (part_.owner[t] < 0 && my_rank == 0))
        for (int k = 0; k < nodes[t].ns; ++k) own[sym.perm_[nodes[t].first + k]] = 1;
    d_own_orig_ = upload(own);
    std::vector<int> kind(n_, 0), top_unk;
    for (int t = 0; t < nnodes_; ++t)
      for (int k = 0; k < nodes[t].ns; ++k) {
        const int i = sym.perm_[nodes[t].first + k];
        kind[i] = part_.owner[t] < 0 ? 2 : (part_.owner[t] == my_rank ? 1 : 0);
        if (part_.owner[t] < 0) top_unk.push_back(i);
      }
    std::sort(top_unk.begin(), top_unk.end());
    d_kind_orig_ = upload(kind);
    ntop_unk_ = (int)top_unk.size();
    d_top_unk_ = upload(top_unk);
    vals_local_ = sym.rank_aligned(part_.world);
    if (vals_local_) {
      const std::vector<int> top_idx = sym.top_value_indices(part_);
      ntop_vals_ = (int)top_idx.size();
      d_top_idx_ = upload(top_idx);
    }
    ck(hipMalloc((void**)&d_xchg_, std::max<long long>(xchg_doubles_ + ntop_vals_, 1) * sizeof(double)), "hipMalloc xchg");
    allocs_.push_back(d_xchg_);
    ck(hipMalloc((void**)&d_xsol_, (size_t)(n_ + 1) * sizeof(double)), "hipMalloc xsol");
    allocs_.push_back(d_xsol_);
  }
  ck(hipMalloc((void**)&d_fronts_, std::max<long long>(total_front_, 1) * sizeof(double)), "hipMalloc fronts");
  allocs_.push_back(d_fronts_);
  // the mirrored-L half of every front is written before it is read, but never leave it uninitialised
  ck(hipMemset(d_fronts_, 0, std::max<long long>(total_front_, 1) * sizeof(double)), "memset fronts");
  ck(hipMalloc((void**)&d_linv_, std::max<long long>(loff, 1) * sizeof(double)), "hipMalloc linv");
  allocs_.push_back(d_linv_);
  ck(hipMalloc((void**)&d_rect_, std::max(n_, 1) * sizeof(double)), "hipMalloc rect");
  allocs_.push_back(d_rect_);
  ck(hipMalloc((void**)&d_y_, std::max(n_, 1) * sizeof(double)), "hipMalloc y");
  allocs_.push_back(d_y_);
  ck(hipMalloc((void**)&d_fail_, sizeof(int)), "hipMalloc flag");
  allocs_.push_back(d_fail_);
  ck(hipMemset(d_fail_, 0, sizeof(int)), "memset");
  if (std::getenv("MGB_CHOL_PROF")) {
    ck(hipMalloc((void**)&d_prof_, (size_t)kProfSlots * launches_ * sizeof(long long)), "hipMalloc prof");
    allocs_.push_back(d_prof_);
#ifdef MGB_PROF_PER_WG
    {
      long long* wg = nullptr;
      ck(hipMalloc((void**)&wg, (size_t)launches_ * kProfMaxWg * 2 * sizeof(long long)), "hipMalloc wgprof");
      allocs_.push_back(wg);
      ck(hipMemset(wg, 0, (size_t)launches_ * kProfMaxWg * 2 * sizeof(long long)), "memset wgprof");
      ck(hipMemcpyToSymbol(HIP_SYMBOL(g_wgprof), &wg, sizeof(wg)), "symbol wgprof");
      ck(hipMemcpyToSymbol(HIP_SYMBOL(g_prof_base), &d_prof_, sizeof(d_prof_)), "symbol prof base");
      d_wgprof_ = wg;
    }
#endif
    {
      std::vector<long long> init((size_t)kProfSlots * launches_, 0LL);
      for (int q = 0; q < launches_; ++q) init[(size_t)kProfSlots * q + 8] = -1LL;
      ck(hipMemcpy(d_prof_, init.data(), init.size() * sizeof(long long), hipMemcpyHostToDevice), "init prof");
    }
  }
  // per device (function attributes do not carry over to another GPU of the same process): set on every build
  ck(hipFuncSetAttribute((const void*)backward_kernel<256>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024), "attr");
  ck(hipFuncSetAttribute((const void*)backward_kernel<1024>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024), "attr");
  ck(hipFuncSetAttribute((const void*)backward_rect_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024), "attr");
  ck(hipFuncSetAttribute((const void*)front_leaf_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024), "attr");
  ck(hipFuncSetAttribute((const void*)front_step2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kStep2Lds), "attr");
  ck(hipFuncSetAttribute((const void*)front_panel2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kPanel2Lds), "attr");
  ck(hipFuncSetAttribute((const void*)front_update2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kUpdate2Lds), "attr");
}

// The whole chain is launch-bound (37 dependent launches at fem2d L=7), so it is captured once per
// (values, rhs, solution) pointer triple into a hipGraph and replayed with ONE host call per Newton step; the
// event-timed and phase-stamped variants (KernelTimer, MGB_CHOL_PROF) and MGB_CHOL_GRAPH=0 use plain launches.
void GpuChol::factor_solve(hipStream_t st, double* d_vals, const double* d_b, double* d_x, KernelTimer* tm, bool flag_armed,
                           bool values_summed, bool x_local) {
  if (n_ == 0) return;
  static const bool use_graph = [] {
    const char* e = std::getenv("MGB_CHOL_GRAPH");
    return !(e && e[0] == '0');
  }();
  // the pivot flag is re-armed here, outside the captured chain (a memset node replayed from the graph was seen
  // to leave garbage in the flag when another library used the device between replays)
  if (!flag_armed) ck(hipMemsetAsync(d_fail_, 0, sizeof(int), st), "memset flag");
  if (part_.split()) {      // two collectives inside: plain launches, no graph
    factor_solve_split(st, d_vals, d_b, d_x, tm, values_summed, x_local && vals_local_ && !values_summed);
    return;
  }
  if (!use_graph || tm || d_prof_) {
    enqueue(st, d_vals, d_b, d_x, tm);
    return;
  }
  for (const GraphEntry& g : graphs_)
    if (g.vals == d_vals && g.b == d_b && g.x == d_x) {
      ck(hipGraphLaunch(g.exec, st), "hipGraphLaunch");
      return;
    }
  hipGraph_t graph = nullptr;
  ck(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal), "hipStreamBeginCapture");
  try {
    enqueue(st, d_vals, d_b, d_x, nullptr);
  } catch (...) {
    (void)hipStreamEndCapture(st, &graph);
    if (graph) (void)hipGraphDestroy(graph);
    throw;
  }
  ck(hipStreamEndCapture(st, &graph), "hipStreamEndCapture");
  GraphEntry ge{d_vals, d_b, d_x, nullptr};
  ck(hipGraphInstantiate(&ge.exec, graph, nullptr, nullptr, 0), "hipGraphInstantiate");
  (void)hipGraphDestroy(graph);
  graphs_.push_back(ge);
  ck(hipGraphLaunch(ge.exec, st), "hipGraphLaunch");
}

void GpuChol::enqueue_forward(hipStream_t st, const std::vector<HeightPlan>& plan, const double* d_vals, const double* d_b,
                              KernelTimer* tm, int& nprof) {
  for (const HeightPlan& hp : plan) {
    if (hp.leaf) {
      if (tm) tm->begin(st, KC_CHOL_SINGLE, hp.start_bytes);
      const size_t lds = ((size_t)(hp.max_nf + 1) * hp.max_nf - (size_t)hp.max_nf * (hp.max_nf - 1) / 2) * sizeof(double);
      hipLaunchKernelGGL(front_leaf_kernel, dim3(hp.nodes.cnt), dim3(TB), lds, st, d_hnodes_ + hp.nodes.ofs, d_lists_ + hp.nodes.ofs, d_asm_src_,
                         d_asm_pos_, d_vals, d_perm_, d_b, d_fronts_, d_linv_, d_fail_,
                         d_prof_ ? d_prof_ + kProfSlots * (nprof++) : nullptr);
      if (tm) tm->end(st);
      continue;
    }
    if (hp.single) {
      if (tm) tm->begin(st, KC_CHOL_SINGLE, hp.start_bytes + hp.step_bytes[0]);
      static const int dense_tiles = [] {      // MGB_CHOL_DENSE_TILES: launches above this many tiles use the three-per-CU variant
        const char* e = std::getenv("MGB_CHOL_DENSE_TILES");
        return e ? std::atoi(e) : 512;      // what two tiles per CU hold at once
      }();
      const bool dense = hp.single_tiles.cnt > dense_tiles;
      hipLaunchKernelGGL(dense ? (hp.narrow ? front_single_dense_kernel<true> : front_single_dense_kernel<false>)
                               : (hp.narrow ? front_single_kernel<true> : front_single_kernel<false>),
                         dim3(hp.single_tiles.cnt), dim3(TB), 0, st, d_singles_ + hp.single_tiles.ofs, d_pinv_,
                         d_asm_src_, d_asm_pos_, d_vals, d_perm_, d_b, d_fronts_, d_fronts_, d_linv_, d_fail_,
                         d_prof_ ? d_prof_ + kProfSlots * (nprof++) : nullptr);
      if (tm) tm->end(st);
      continue;
    }
    if (tm) tm->begin(st, KC_CHOL_START, hp.start_bytes);
    hipLaunchKernelGGL(front_start_kernel, dim3(hp.start.cnt), dim3(TB), 0, st, d_nodes_, d_start_ + hp.start.ofs, d_pinv_,
                       d_asm_src_, d_asm_pos_, d_vals, d_perm_, d_b, d_fronts_, d_fronts_, d_linv_, d_fail_,
                       d_prof_ ? d_prof_ + kProfSlots * (nprof++) : nullptr, start_pivot_ ? 1 : 0);
    if (tm) tm->end(st);
    for (size_t q = 0; q < hp.step.size(); ++q) {
      if (tm) tm->begin(st, KC_CHOL_STEP, hp.step_bytes[q]);
      if (hp.step_pair[q] == 2)
        hipLaunchKernelGGL(front_panel2_kernel, dim3(hp.step[q].cnt), dim3(TB), kPanel2Lds, st, d_tiles_ + hp.step[q].ofs, hp.step_p[q],
                           hp.step_npiv[q], d_fronts_, d_linv_, d_linv_, d_fail_, d_prof_ ? d_prof_ + kProfSlots * (nprof++) : nullptr);
      else if (hp.step_pair[q] == 3)
        hipLaunchKernelGGL(front_update2_kernel, dim3(hp.step[q].cnt), dim3(TB), kUpdate2Lds, st, d_tiles_ + hp.step[q].ofs, hp.step_p[q],
                           0, d_fronts_, d_linv_, d_linv_, d_fail_, d_prof_ ? d_prof_ + kProfSlots * (nprof++) : nullptr);
      else if (hp.step_pair[q])
        hipLaunchKernelGGL(front_step2_kernel, dim3(hp.step[q].cnt), dim3(TB), kStep2Lds, st, d_tiles_ + hp.step[q].ofs, hp.step_p[q],
                           hp.step_npiv[q], d_fronts_, d_linv_, d_linv_, d_fail_, d_prof_ ? d_prof_ + kProfSlots * (nprof++) : nullptr);
      else
        hipLaunchKernelGGL(front_step_kernel, dim3(hp.step[q].cnt), dim3(TB), 0, st, d_tiles_ + hp.step[q].ofs, hp.step_p[q],
                           hp.step_npiv[q], d_fronts_, d_linv_, d_linv_, d_fail_, d_prof_ ? d_prof_ + kProfSlots * (nprof++) : nullptr);
      if (tm) tm->end(st);
    }
  }
}

void GpuChol::enqueue_backward(hipStream_t st, const std::vector<HeightPlan>& plan, double* d_x, KernelTimer* tm, int* nprof) {
  for (int h = (int)plan.size() - 1; h >= 0; --h) {
    const HeightPlan& hp = plan[h];
    const int use_rect = hp.rect.cnt ? 1 : 0;
    if (use_rect) {
      if (tm) tm->begin(st, KC_CHOL_BWD_RECT, hp.rect_bytes);
      hipLaunchKernelGGL(backward_rect_kernel, dim3(hp.rect.cnt), dim3(RT), (size_t)(hp.max_nf + RT) * sizeof(double), st,
                         d_rnodes_ + hp.rect.ofs, d_rectjobs_ + hp.rect.ofs, d_bdry_, d_fronts_, d_y_, d_rect_);
      if (tm) tm->end(st);
    }
    if (tm) tm->begin(st, KC_CHOL_BWD, hp.tri_bytes + (use_rect ? 0.0 : hp.rect_bytes));
    if (hp.max_nf > 384)
      hipLaunchKernelGGL(backward_kernel<1024>, dim3(hp.nodes.cnt), dim3(1024), (size_t)(hp.max_nf + 1024 + PB) * sizeof(double),
                         st, d_hnodes_ + hp.nodes.ofs, d_lists_ + hp.nodes.ofs, d_bdry_, d_fronts_, d_linv_, d_perm_, d_rect_, use_rect, d_y_,
                         d_x, (d_prof_ && nprof) ? d_prof_ + kProfSlots * ((*nprof)++) : nullptr);
    else
      hipLaunchKernelGGL(backward_kernel<256>, dim3(hp.nodes.cnt), dim3(256), (size_t)(hp.max_nf + 256 + PB) * sizeof(double),
                         st, d_hnodes_ + hp.nodes.ofs, d_lists_ + hp.nodes.ofs, d_bdry_, d_fronts_, d_linv_, d_perm_, d_rect_, use_rect, d_y_,
                         d_x, (d_prof_ && nprof) ? d_prof_ + kProfSlots * ((*nprof)++) : nullptr);
    if (tm) tm->end(st);
  }
}

void GpuChol::enqueue_chain(hipStream_t st, const double* d_vals, const double* d_b, double* d_x) {
  if (part_.split()) throw InternalError("gpuchol: a split factorisation cannot be captured");
  if (n_ == 0) return;
  int nprof = 0;
  enqueue_forward(st, plan_, d_vals, d_b, nullptr, nprof);
  enqueue_backward(st, plan_, d_x, nullptr);
  ck(hipGetLastError(), "chain launches");
}

void GpuChol::factor_solve_split(hipStream_t st, double* d_vals, const double* d_b, double* d_x, KernelTimer* tm, bool values_summed,
                                 bool x_local) {
  int nprof = 0;
  const bool ride = vals_local_ && !values_summed;
  const int ntop = ride ? ntop_vals_ : 0;
  enqueue_forward(st, plan_, d_vals, d_b, tm, nprof);                    // this rank's subtree
  const dim3 xg(nroots_, std::max(1, std::min(64, max_root_nb_)));
  hipLaunchKernelGGL(schur_pack_kernel, xg, dim3(256), 0, st, d_roots_, d_fronts_, d_xchg_);
  const int vgrid = std::max(1, std::min(1024, (ntop_vals_ + 255) / 256));
  if (ride)      // this rank's partial sums of the top nodes' matrix entries ride along
    hipLaunchKernelGGL(vals_pack_kernel, dim3(vgrid), dim3(256), 0, st, ntop_vals_, d_top_idx_, d_vals, d_xchg_ + xchg_doubles_);
  ck(hipGetLastError(), "schur pack");
  ctx_->allreduce_sum(d_xchg_, xchg_doubles_ + ntop);                    // every subtree root's Schur complement, everywhere
  hipLaunchKernelGGL(schur_unpack_kernel, xg, dim3(256), 0, st, d_roots_, d_xchg_, d_fronts_);
  if (ride)
    hipLaunchKernelGGL(vals_unpack_kernel, dim3(vgrid), dim3(256), 0, st, ntop_vals_, d_top_idx_, d_xchg_ + xchg_doubles_, d_vals);
  enqueue_forward(st, plan_top_, d_vals, d_b, tm, nprof);                // replicated top: same arithmetic on every rank
  enqueue_backward(st, plan_top_, d_x, tm);
  enqueue_backward(st, plan_, d_x, tm);
  const int grid = std::min(2048, (n_ + 256) / 256);
  if (x_local) {      // every rank holds what its rows need: its subtree's unknowns and the replicated top
    hipLaunchKernelGGL(x_local_kernel, dim3(grid), dim3(256), 0, st, n_, d_kind_orig_, d_x);
    ck(hipGetLastError(), "x_local");
    return;
  }
  hipLaunchKernelGGL(xsol_pack_kernel, dim3(grid), dim3(256), 0, st, n_, d_own_orig_, d_x, d_fail_, d_xsol_);
  ck(hipGetLastError(), "xsol pack");
  ctx_->allreduce_sum(d_xsol_, (long long)n_ + 1);                       // x (one contributor per unknown) + pivot flag
  hipLaunchKernelGGL(xsol_unpack_kernel, dim3(grid), dim3(256), 0, st, n_, d_xsol_, d_x, d_fail_);
  ck(hipGetLastError(), "factor_solve_split launches");
}

void GpuChol::enqueue(hipStream_t st, const double* d_vals, const double* d_b, double* d_x, KernelTimer* tm) {
  int nprof = 0;
  enqueue_forward(st, plan_, d_vals, d_b, tm, nprof);
  enqueue_backward(st, plan_, d_x, tm, &nprof);
  ck(hipGetLastError(), "factor_solve launches");
  if (d_prof_) {      // debugging aid: phase stamps of workgroup 0 of every factorisation launch, in units of 10 ns
    ck(hipStreamSynchronize(st), "prof sync");
    std::vector<long long> hprof((size_t)kProfSlots * nprof);
    ck(hipMemcpy(hprof.data(), d_prof_, hprof.size() * sizeof(long long), hipMemcpyDeviceToHost), "prof D2H");
    std::fprintf(stderr, "[mgb chol prof] launch: load trsm update store sync Dwrite factor tail (us)\n");
    for (int q = 0; q < nprof; ++q) {
      const long long* v = hprof.data() + kProfSlots * q;
      std::fprintf(stderr, "[mgb chol prof] %3d:", q);
      for (int k = 1; k < 8; ++k) std::fprintf(stderr, " %6.2f", (v[k] && v[k - 1]) ? (v[k] - v[k - 1]) * 0.01 : 0.0);
      // all workgroups: first start -> last end, and the time since the previous launch's last end (launch boundary)
      const long long* pv = q ? hprof.data() + kProfSlots * (q - 1) : nullptr;
      const long long end_q = (long long)((unsigned long long)v[9] >> 16), end_p = pv ? (long long)((unsigned long long)pv[9] >> 16) : 0;
      std::fprintf(stderr, "  | all wgs %6.2f (last: wg %5d)  since prev end %6.2f\n", (end_q - v[8]) * 0.01,
                   (int)((unsigned long long)v[9] & 65535ull), pv ? (v[8] - end_p) * 0.01 : 0.0);
    }
#ifdef MGB_PROF_PER_WG
    if (d_wgprof_) {
      std::vector<long long> wg((size_t)nprof * kProfMaxWg * 2);
      ck(hipMemcpy(wg.data(), d_wgprof_, wg.size() * sizeof(long long), hipMemcpyDeviceToHost), "wgprof D2H");
      std::fprintf(stderr, "[mgb chol wgs] launch: workgroups | start of the last one after the first | duration min / median / max | span | slowest: wg(start, duration) (us)\n");
      for (int q = 0; q < nprof; ++q) {
        std::vector<std::array<double, 3>> v;      // start, duration, id
        long long t0 = -1;
        for (int b = 0; b < kProfMaxWg; ++b) {
          const long long st0 = wg[((size_t)q * kProfMaxWg + b) * 2], en = wg[((size_t)q * kProfMaxWg + b) * 2 + 1];
          if (st0 == 0 || en == 0) continue;
          if (t0 < 0 || st0 < t0) t0 = st0;
        }
        double last_start = 0, span = 0;
        for (int b = 0; b < kProfMaxWg; ++b) {
          const long long st0 = wg[((size_t)q * kProfMaxWg + b) * 2], en = wg[((size_t)q * kProfMaxWg + b) * 2 + 1];
          if (st0 == 0 || en == 0) continue;
          v.push_back({(st0 - t0) * 0.01, (en - st0) * 0.01, (double)b});
          last_start = std::max(last_start, (st0 - t0) * 0.01);
          span = std::max(span, (en - t0) * 0.01);
        }
        if (v.empty()) continue;
        std::sort(v.begin(), v.end(), [](const auto& a, const auto& b2) { return a[1] < b2[1]; });
        std::fprintf(stderr, "[mgb chol wgs] %3d: %5zu | %6.2f | %6.2f %6.2f %6.2f | %6.2f |", q, v.size(), last_start, v.front()[1],
                     v[v.size() / 2][1], v.back()[1], span);
        for (size_t k = v.size() > 4 ? v.size() - 4 : 0; k < v.size(); ++k)
          std::fprintf(stderr, " %d(%.2f, %.2f)", (int)v[k][2], v[k][0], v[k][1]);
        // the last finishers
        std::sort(v.begin(), v.end(), [](const auto& a, const auto& b2) { return a[0] + a[1] < b2[0] + b2[1]; });
        std::fprintf(stderr, " | last to end:");
        for (size_t k = v.size() > 3 ? v.size() - 3 : 0; k < v.size(); ++k)
          std::fprintf(stderr, " %d(%.2f, %.2f)", (int)v[k][2], v[k][0], v[k][1]);
        std::fprintf(stderr, "\n");
      }
      ck(hipMemset(d_wgprof_, 0, wg.size() * sizeof(long long)), "wgprof reset");
    }
#endif
    for (int q = 0; q < nprof; ++q) {
      std::fill(hprof.begin() + (size_t)kProfSlots * q, hprof.begin() + (size_t)kProfSlots * (q + 1), 0LL);
      hprof[(size_t)kProfSlots * q + 8] = -1LL;      // all ones: atomicMin target (unsigned)
    }
    ck(hipMemcpy(d_prof_, hprof.data(), hprof.size() * sizeof(long long), hipMemcpyHostToDevice), "prof reset");
  }
}

}  // namespace mgb
