// gfx950 kernels + host schedule of the device multifrontal Cholesky (see gpuchol.hpp).
//
// Roofline: the trailing update is the only compute-shaped kernel (fp64 rank-32 updates on 32x32 tiles,
// ~1.5 GFLOP per fine-level factorisation at fem2d L=7); everything else is latency bound (a dependent
// chain of ~150 short launches per factorisation).  Fronts (225 MB at L=7) stream through L2/Infinity
// Cache; no MFMA (fp64 dense work is small and triangular/ragged).
#include "gpuchol.hpp"

#include "amg.hpp"

#include <algorithm>
#include <cmath>
#include <stdexcept>
#include <string>

namespace mgb {

namespace {

constexpr int PB = 32;     // panel width
constexpr int TB = 256;    // threads per workgroup

void ck(hipError_t e, const char* what) {
  if (e != hipSuccess) throw std::runtime_error(std::string("HIP error in gpuchol ") + what + ": " + hipGetErrorString(e));
}

__global__ __launch_bounds__(TB) void scatter_kernel(int n, const int* __restrict__ src,
                                                      const long long* __restrict__ dst,
                                                      const double* __restrict__ vals, double* fronts) {
  for (long long k = (long long)blockIdx.x * TB + threadIdx.x; k < n; k += (long long)gridDim.x * TB)
    fronts[dst[k]] = vals[src[k]];
}

// broadcast lane `lane` (compile-time constant) of a double through SGPRs
__device__ inline double readlane_f64(double v, int lane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}

// parent front += child's Schur complement.  One workgroup per (child, chunk of 8 boundary columns); the two
// child slots of a height are separate launches, so siblings never add into the same entry concurrently.
constexpr int EA_COLS = 8;
__global__ __launch_bounds__(TB) void extend_add_kernel(const GNode* __restrict__ nodes, const GTile* __restrict__ list,
                                                         const int* __restrict__ ea_all, double* fronts) {
  const GTile job = list[blockIdx.x];
  const GNode c = nodes[job.node];
  const GNode p = nodes[c.parent];
  const int nb = c.nf - c.ns;
  const int* ea = ea_all + c.bofs;
  const double* Fc = fronts + c.off;
  double* Fp = fronts + p.off;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b0 = job.ti * EA_COLS, b1 = min(nb, b0 + EA_COLS);
  for (int b = b0 + wave; b < b1; b += TB / 64) {
    const long long pc = (long long)p.nf * ea[b];
    const double* col = Fc + (long long)c.nf * (c.ns + b) + c.ns;
    for (int a = b + lane; a < nb; a += 64) Fp[pc + ea[a]] += col[a];
  }
}

// Panel p of every listed front, rows [k1 + 256*slice, ...): every workgroup re-derives the Cholesky factor
// of the ORIGINAL 32x32 diagonal block (wave-level, rows in registers, cross-lane shuffles) and its inverse,
// so the slices of one front run concurrently without reading anything another slice writes; slice 0 stores
// the inverse (both orientations) to the scratch the sweeps read.  L21 = F21 * L11^{-T}.
constexpr int SLICE = TB;
__global__ __launch_bounds__(TB) void panel_factor_kernel(const GNode* __restrict__ nodes, const GTile* __restrict__ list,
                                                           int p, double* fronts, double* linv, int* fail) {
  __shared__ double Ls[PB][PB + 1];
  __shared__ double Is[PB][PB + 1];
  const GTile job = list[blockIdx.x];
  const GNode nd = nodes[job.node];
  const int nf = nd.nf, k0 = p * PB, kw = min(PB, nd.ns - k0), k1 = k0 + kw;
  double* F = fronts + nd.off;
  const int tid = threadIdx.x;
  if (tid < 64) {
    // lane i (< 32) owns row i of the block (padded with the identity beyond kw).  Column k of L is
    // broadcast with constant-lane v_readlane (no LDS round trip); entries right of the diagonal of a row
    // are don't-care (their multiplier l is forced to 0 once k passes the row), so the update is branch-free.
    const int i = tid & 31;
    double a[PB], rdv[PB];
#pragma unroll
    for (int j = 0; j < PB; ++j) {
      double v = (i == j) ? 1.0 : 0.0;
      if (i < kw && j < kw && j <= i) v = F[(long long)nf * (k0 + j) + k0 + i];
      a[j] = v;
    }
    bool bad = false;
#pragma unroll
    for (int k = 0; k < PB; ++k) {
      double akk = readlane_f64(a[k], k);
      if (!(akk > 0.0) || !isfinite(akk)) {
        bad = true;
        akk = 1.0;
      }
      const double rd = rsqrt(akk);          // 1 / L[k][k]
      rdv[k] = rd;
      const double l = (i > k) ? a[k] * rd : ((i == k) ? akk * rd : 0.0);
      a[k] = l;
#pragma unroll
      for (int j = k + 1; j < PB; ++j) a[j] = fma(-l, readlane_f64(l, j), a[j]);
    }
    if (bad && tid == 0) atomicOr(fail, 1);
    if (tid < PB) {
#pragma unroll
      for (int j = 0; j < PB; ++j) Ls[i][j] = (j <= i) ? a[j] : 0.0;
      // column c = i of L^{-1} by forward substitution; L entries are LDS broadcasts of OTHER rows, so
      // make the block visible to the wave first
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    if (tid < PB) {
      const int c = i;
      double x[PB];
#pragma unroll
      for (int r = 0; r < PB; ++r) {
        double s0 = (r == c) ? 1.0 : 0.0, s1 = 0.0;
#pragma unroll
        for (int m = 0; m + 1 < r; m += 2) {
          s0 = fma(-Ls[r][m], x[m], s0);
          s1 = fma(-Ls[r][m + 1], x[m + 1], s1);
        }
        if (r & 1) s0 = fma(-Ls[r][r - 1], x[r - 1], s0);
        x[r] = (s0 + s1) * rdv[r];
      }
#pragma unroll
      for (int r = 0; r < PB; ++r) Is[r][c] = (r >= c) ? x[r] : 0.0;
    }
  }
  __syncthreads();
  if (job.ti == 0) {
    double* lp = linv + nd.loff + (long long)p * 2 * PB * PB;
    for (int idx = tid; idx < PB * PB; idx += TB) {
      const int i = idx % PB, j = idx / PB;
      lp[idx] = Is[i][j];                 // Linv, column-major: lp[i + 32 j] = Linv[i][j]
      lp[PB * PB + idx] = Is[j][i];       // its transpose, column-major
    }
  }
  // rows of this slice: x = f * L11^{-T}, i.e. x_j = sum_{m<=j} f_m Linv[j][m]
  const int i = k1 + job.ti * SLICE + tid;
  if (i < nf) {
    double f[PB];
#pragma unroll
    for (int j = 0; j < PB; ++j) f[j] = (j < kw) ? F[(long long)nf * (k0 + j) + i] : 0.0;
#pragma unroll
    for (int j = 0; j < PB; ++j) {
      if (j < kw) {
        double s = 0.0;
#pragma unroll
        for (int m = 0; m <= j; ++m) s += f[m] * Is[j][m];
        F[(long long)nf * (k0 + j) + i] = s;
      }
    }
  }
}

// C[i,j] -= sum_q L[i,k0+q] L[j,k0+q] on one 32x32 tile of the trailing lower triangle
__global__ __launch_bounds__(TB) void trailing_update_kernel(const GNode* __restrict__ nodes,
                                                              const GTile* __restrict__ tiles, int p, double* fronts) {
  __shared__ double Pi[PB][PB + 1];
  __shared__ double Pj[PB][PB + 1];
  const GTile t = tiles[blockIdx.x];
  const GNode nd = nodes[t.node];
  const int nf = nd.nf, k0 = p * PB, kw = min(PB, nd.ns - k0), k1 = k0 + kw;
  double* F = fronts + nd.off;
  const int r0 = k1 + PB * t.ti, c0 = k1 + PB * t.tj;
  const int tid = threadIdx.x;
  for (int idx = tid; idx < PB * PB; idx += TB) {
    const int r = idx % PB, q = idx / PB;
    Pi[r][q] = (q < kw && r0 + r < nf) ? F[(long long)nf * (k0 + q) + r0 + r] : 0.0;
    Pj[r][q] = (q < kw && c0 + r < nf) ? F[(long long)nf * (k0 + q) + c0 + r] : 0.0;
  }
  __syncthreads();
  const int tx = tid & 15, ty = tid >> 4;
  double a00 = 0, a01 = 0, a10 = 0, a11 = 0;
#pragma unroll 8
  for (int q = 0; q < PB; ++q) {
    const double x0 = Pi[tx][q], x1 = Pi[tx + 16][q], y0 = Pj[ty][q], y1 = Pj[ty + 16][q];
    a00 += x0 * y0;
    a01 += x0 * y1;
    a10 += x1 * y0;
    a11 += x1 * y1;
  }
  const int i0 = r0 + tx, i1 = r0 + tx + 16, j0 = c0 + ty, j1 = c0 + ty + 16;
  if (i0 < nf && j0 < nf && i0 >= j0) F[(long long)nf * j0 + i0] -= a00;
  if (i0 < nf && j1 < nf && i0 >= j1) F[(long long)nf * j1 + i0] -= a01;
  if (i1 < nf && j0 < nf && i1 >= j0) F[(long long)nf * j0 + i1] -= a10;
  if (i1 < nf && j1 < nf && i1 >= j1) F[(long long)nf * j1 + i1] -= a11;
}

__global__ __launch_bounds__(TB) void gather_perm_kernel(int n, const int* __restrict__ perm,
                                                          const double* __restrict__ b, double* y) {
  for (long long i = (long long)blockIdx.x * TB + threadIdx.x; i < n; i += (long long)gridDim.x * TB) y[i] = b[perm[i]];
}

__global__ __launch_bounds__(TB) void scatter_perm_kernel(int n, const int* __restrict__ perm,
                                                           const double* __restrict__ y, double* x) {
  for (long long i = (long long)blockIdx.x * TB + threadIdx.x; i < n; i += (long long)gridDim.x * TB) x[perm[i]] = y[i];
}

// Forward sweep L u = b of one height: front-local vector u = [own | bdry] in LDS; children's boundary
// parts are pulled in slot order, the own part is solved panel by panel with the stored block inverses,
// then the boundary part is updated in one pass (u_bdry -= L21 u_own).
template <int NT>
__global__ __launch_bounds__(NT) void forward_kernel(const GNode* __restrict__ nodes, const int* __restrict__ list,
                                                      const int* __restrict__ ea_all, const double* __restrict__ fronts,
                                                      const double* __restrict__ linv, double* y, double* work) {
  extern __shared__ double u[];
  __shared__ double tmp[PB];
  const GNode nd = nodes[list[blockIdx.x]];
  const int nf = nd.nf, ns = nd.ns, nb = nf - ns, tid = threadIdx.x;
  const double* F = fronts + nd.off;
  for (int i = tid; i < nf; i += NT) u[i] = (i < ns) ? y[nd.first + i] : 0.0;
  __syncthreads();
  for (int s = 0; s < 2; ++s) {
    if (nd.child[s] >= 0) {
      const GNode c = nodes[nd.child[s]];
      const int cnb = c.nf - c.ns;
      const int* ea = ea_all + c.bofs;
      const double* wc = work + c.woff + c.ns;
      for (int i = tid; i < cnb; i += NT) u[ea[i]] += wc[i];
    }
    __syncthreads();
  }
  for (int k0 = 0; k0 < ns; k0 += PB) {
    const int kw = min(PB, ns - k0), k1 = k0 + kw;
    if (tid < kw) {
      const int j = tid;
      const double* lp = linv + nd.loff + (long long)(k0 / PB) * 2 * PB * PB;    // Linv[j][m] at lp[j + 32 m]
      double s = 0.0;
      for (int m = 0; m <= j; ++m) s += lp[j + PB * m] * u[k0 + m];
      tmp[j] = s;
    }
    __syncthreads();
    if (tid < kw) u[k0 + tid] = tmp[tid];
    __syncthreads();
    for (int i = k1 + tid; i < ns; i += NT) {
      double d = 0.0;
      for (int j = 0; j < kw; ++j) d += F[(long long)nf * (k0 + j) + i] * u[k0 + j];
      u[i] -= d;
    }
    __syncthreads();
  }
  for (int i = tid; i < ns; i += NT) y[nd.first + i] = u[i];
  double* w = work + nd.woff + ns;
  for (int i = tid; i < nb; i += NT) {
    const double* row = F + ns + i;
    double d0 = 0.0, d1 = 0.0;
    int j = 0;
    for (; j + 1 < ns; j += 2) {
      d0 += row[(long long)nf * j] * u[j];
      d1 += row[(long long)nf * (j + 1)] * u[j + 1];
    }
    if (j < ns) d0 += row[(long long)nf * j] * u[j];
    w[i] = u[ns + i] - (d0 + d1);
  }
}

// Backward sweep L' x = u of one height (heights descending): ancestors' entries of x are final.
template <int NT>
__global__ __launch_bounds__(NT) void backward_kernel(const GNode* __restrict__ nodes, const int* __restrict__ list,
                                                       const int* __restrict__ bdry_all,
                                                       const double* __restrict__ fronts,
                                                       const double* __restrict__ linv, double* y) {
  extern __shared__ double u[];      // [0,ns): rhs -> solution ; [ns,nf): x of the boundary dofs
  __shared__ double tmp[PB];
  const GNode nd = nodes[list[blockIdx.x]];
  const int nf = nd.nf, ns = nd.ns, nb = nf - ns, tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const double* F = fronts + nd.off;
  const int* bd = bdry_all + nd.bofs;
  for (int i = tid; i < nf; i += NT) u[i] = (i < ns) ? y[nd.first + i] : y[bd[i - ns]];
  __syncthreads();
  // u_own -= L21' x_bdry : column j of L21 is contiguous
  for (int j = wave; j < ns; j += NT / 64) {
    const double* col = F + (long long)nf * j + ns;
    double s = 0.0;
    for (int i = lane; i < nb; i += 64) s += col[i] * u[ns + i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
    if (lane == 0) u[j] -= s;
  }
  __syncthreads();
  const int npanel = (ns + PB - 1) / PB;
  for (int pp = npanel - 1; pp >= 0; --pp) {
    const int k0 = pp * PB, kw = min(PB, ns - k0), k1 = k0 + kw;
    // u_p -= L[k1:ns, panel]' x[k1:ns]
    for (int c = wave; c < kw; c += NT / 64) {
      const double* col = F + (long long)nf * (k0 + c);
      double s = 0.0;
      for (int i = k1 + lane; i < ns; i += 64) s += col[i] * u[i];
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
      if (lane == 0) u[k0 + c] -= s;
    }
    __syncthreads();
    if (tid < kw) {
      const int c = tid;     // x_c = sum_{m>=c} Linv[m][c] u_m ; transpose block: lt[c + 32 m] = Linv[m][c]
      const double* lt = linv + nd.loff + (long long)pp * 2 * PB * PB + PB * PB;
      double s = 0.0;
      for (int m = c; m < kw; ++m) s += lt[c + PB * m] * u[k0 + m];
      tmp[c] = s;
    }
    __syncthreads();
    if (tid < kw) u[k0 + tid] = tmp[tid];
    __syncthreads();
  }
  for (int i = tid; i < ns; i += NT) y[nd.first + i] = u[i];
}

inline int blocks_for(long long n) {
  long long b = (n + TB - 1) / TB;
  return (int)std::max<long long>(1, std::min<long long>(b, 2048));
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
  for (void* p : allocs_) (void)hipFree(p);
}

void GpuChol::build(const MfChol& sym) {
  n_ = sym.n_;
  nnodes_ = (int)sym.nodes_.size();
  flops_ = sym.flops_;
  std::vector<GNode> nodes(nnodes_);
  std::vector<int> bdry_all, ea_all, height(nnodes_, 0);
  long long off = 0, woff = 0, loff = 0;
  max_nf_ = 0;
  for (int t = 0; t < nnodes_; ++t) {
    const auto& nd = sym.nodes_[t];
    GNode& g = nodes[t];
    g.off = off;
    g.woff = woff;
    g.loff = loff;
    loff += (long long)((nd.ns + PB - 1) / PB) * 2 * PB * PB;
    g.nf = nd.nf();
    g.ns = nd.ns;
    g.first = nd.first;
    g.parent = nd.parent;
    g.bofs = (int)bdry_all.size();
    g.child[0] = nd.children.size() > 0 ? nd.children[0] : -1;
    g.child[1] = nd.children.size() > 1 ? nd.children[1] : -1;
    if (nd.children.size() > 2) throw std::runtime_error("gpuchol: elimination tree is not binary");
    bdry_all.insert(bdry_all.end(), nd.bdry.begin(), nd.bdry.end());
    ea_all.insert(ea_all.end(), nd.ea.begin(), nd.ea.end());
    if (nd.parent >= 0 && nd.ea.size() != nd.bdry.size()) throw std::runtime_error("gpuchol: ea/bdry size mismatch");
    if (nd.parent < 0) ea_all.resize(bdry_all.size(), 0);
    off += (long long)g.nf * g.nf;
    woff += g.nf;
    max_nf_ = std::max(max_nf_, g.nf);
    for (int c : nd.children) height[t] = std::max(height[t], height[c] + 1);   // postorder: children first
  }
  if (off != (long long)sym.fronts_.size()) throw std::runtime_error("gpuchol: front size mismatch");
  total_front_ = off;
  total_w_ = woff;
  nheights_ = nnodes_ ? *std::max_element(height.begin(), height.end()) + 1 : 0;
  // assembly map
  std::vector<int> asrc;
  std::vector<long long> adst;
  for (int t = 0; t < nnodes_; ++t)
    for (size_t q = 0; q < sym.a_idx_[t].size(); ++q) {
      asrc.push_back(sym.a_idx_[t][q]);
      adst.push_back(nodes[t].off + sym.a_pos_[t][q]);
    }
  nasm_ = (int)asrc.size();
  // schedule
  std::vector<int> lists;
  std::vector<GTile> tiles;
  plan_.assign(nheights_, HeightPlan());
  launches_ = 3;
  for (int h = 0; h < nheights_; ++h) {
    HeightPlan& hp = plan_[h];
    hp.nodes.ofs = (int)lists.size();
    int max_ns = 0;
    hp.max_nf = 0;
    std::vector<int> mine;
    for (int t = 0; t < nnodes_; ++t)
      if (height[t] == h) {
        mine.push_back(t);
        max_ns = std::max(max_ns, nodes[t].ns);
        hp.max_nf = std::max(hp.max_nf, nodes[t].nf);
      }
    lists.insert(lists.end(), mine.begin(), mine.end());
    hp.nodes.cnt = (int)mine.size();
    hp.sweep_bytes = 0;
    for (int t : mine) hp.sweep_bytes += ((double)nodes[t].nf * nodes[t].ns - 0.5 * nodes[t].ns * nodes[t].ns) * 8 + nodes[t].nf * 24.0;
    for (int s = 0; s < 2; ++s) {
      hp.ea[s].ofs = (int)tiles.size();
      hp.ea_bytes[s] = 0;
      for (int t : mine) {
        const int c = nodes[t].child[s];
        if (c < 0) continue;
        const int nb = nodes[c].nf - nodes[c].ns;
        hp.ea_bytes[s] += 0.5 * nb * nb * 24.0;     // read child entry, read+write parent entry
        for (int ch = 0; ch * EA_COLS < nb; ++ch) tiles.push_back({c, (short)ch, 0});
      }
      hp.ea[s].cnt = (int)tiles.size() - hp.ea[s].ofs;
      if (hp.ea[s].cnt) launches_++;
    }
    const int npanel = (max_ns + PB - 1) / PB;
    for (int p = 0; p < npanel; ++p) {
      Range rn{(int)tiles.size(), 0};
      double pbytes = 0;
      for (int t : mine)
        if (nodes[t].ns > p * PB) {
          const int k1 = std::min(nodes[t].ns, (p + 1) * PB);
          pbytes += ((double)(nodes[t].nf - p * PB) * (k1 - p * PB)) * 16.0;   // panel read + written once
          const int nsl = std::max(1, (nodes[t].nf - k1 + SLICE - 1) / SLICE);
          for (int sl = 0; sl < nsl; ++sl) tiles.push_back({t, (short)sl, 0});
        }
      rn.cnt = (int)tiles.size() - rn.ofs;
      Range rt{(int)tiles.size(), 0};
      for (int t : mine)
        if (nodes[t].ns > p * PB) {
          const int k1 = std::min(nodes[t].ns, (p + 1) * PB);
          const int T = (nodes[t].nf - k1 + PB - 1) / PB;
          if (T > 30000) throw std::runtime_error("gpuchol: front too large for tile index");
          for (int ti = 0; ti < T; ++ti)
            for (int tj = 0; tj <= ti; ++tj) tiles.push_back({t, (short)ti, (short)tj});
        }
      rt.cnt = (int)tiles.size() - rt.ofs;
      hp.panel_nodes.push_back(rn);
      hp.panel_bytes.push_back(pbytes);
      hp.panel_tiles.push_back(rt);
      launches_ += 1 + (rt.cnt ? 1 : 0);
    }
  }
  if ((size_t)max_nf_ * 8 > 150 * 1024) throw std::runtime_error("gpuchol: front exceeds the LDS budget of the sweeps");
  d_nodes_ = upload(nodes);
  d_perm_ = upload(sym.perm_);
  d_bdry_ = upload(bdry_all);
  d_ea_ = upload(ea_all);
  d_asm_src_ = upload(asrc);
  d_asm_dst_ = upload(adst);
  d_lists_ = upload(lists);
  d_tiles_ = upload(tiles);
  ck(hipMalloc((void**)&d_fronts_, std::max<long long>(total_front_, 1) * sizeof(double)), "hipMalloc fronts");
  allocs_.push_back(d_fronts_);
  ck(hipMalloc((void**)&d_linv_, std::max<long long>(loff, 1) * sizeof(double)), "hipMalloc linv");
  allocs_.push_back(d_linv_);
  ck(hipMalloc((void**)&d_work_, std::max<long long>(total_w_, 1) * sizeof(double)), "hipMalloc work");
  allocs_.push_back(d_work_);
  ck(hipMalloc((void**)&d_y_, std::max(n_, 1) * sizeof(double)), "hipMalloc y");
  allocs_.push_back(d_y_);
  ck(hipMalloc((void**)&d_fail_, sizeof(int)), "hipMalloc flag");
  allocs_.push_back(d_fail_);
  ck(hipMemset(d_fail_, 0, sizeof(int)), "memset");
  static bool attr_done = false;
  if (!attr_done) {
    ck(hipFuncSetAttribute((const void*)forward_kernel<256>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024), "attr");
    ck(hipFuncSetAttribute((const void*)backward_kernel<256>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024), "attr");
    ck(hipFuncSetAttribute((const void*)forward_kernel<1024>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024), "attr");
    ck(hipFuncSetAttribute((const void*)backward_kernel<1024>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024), "attr");
    attr_done = true;
  }
}

void GpuChol::factor(hipStream_t st, const double* d_vals, KernelTimer* tm) {
  if (n_ == 0) return;
  ck(hipMemsetAsync(d_fronts_, 0, total_front_ * sizeof(double), st), "memset fronts");
  ck(hipMemsetAsync(d_fail_, 0, sizeof(int), st), "memset flag");
  hipLaunchKernelGGL(scatter_kernel, dim3(blocks_for(nasm_)), dim3(TB), 0, st, nasm_, d_asm_src_, d_asm_dst_, d_vals,
                     d_fronts_);
  for (int h = 0; h < nheights_; ++h) {
    const HeightPlan& hp = plan_[h];
    for (int s = 0; s < 2; ++s)
      if (hp.ea[s].cnt) {
        if (tm) tm->begin(st, KC_CHOL_EXTEND, hp.ea_bytes[s]);
        hipLaunchKernelGGL(extend_add_kernel, dim3(hp.ea[s].cnt), dim3(TB), 0, st, d_nodes_, d_tiles_ + hp.ea[s].ofs,
                           d_ea_, d_fronts_);
        if (tm) tm->end(st);
      }
    for (size_t p = 0; p < hp.panel_nodes.size(); ++p) {
      if (tm) tm->begin(st, KC_CHOL_PANEL, hp.panel_bytes[p]);
      hipLaunchKernelGGL(panel_factor_kernel, dim3(hp.panel_nodes[p].cnt), dim3(TB), 0, st, d_nodes_,
                         d_tiles_ + hp.panel_nodes[p].ofs, (int)p, d_fronts_, d_linv_, d_fail_);
      if (tm) tm->end(st);
      if (hp.panel_tiles[p].cnt) {
        if (tm) tm->begin(st, KC_CHOL_TRAIL, hp.panel_tiles[p].cnt * 32768.0);   // 2 panel tiles + C read/write
        hipLaunchKernelGGL(trailing_update_kernel, dim3(hp.panel_tiles[p].cnt), dim3(TB), 0, st, d_nodes_,
                           d_tiles_ + hp.panel_tiles[p].ofs, (int)p, d_fronts_);
        if (tm) tm->end(st);
      }
    }
  }
  ck(hipGetLastError(), "factor launches");
}

void GpuChol::solve(hipStream_t st, const double* d_b, double* d_x, KernelTimer* tm) {
  if (n_ == 0) return;
  hipLaunchKernelGGL(gather_perm_kernel, dim3(blocks_for(n_)), dim3(TB), 0, st, n_, d_perm_, d_b, d_y_);
  for (int h = 0; h < nheights_; ++h) {
    const HeightPlan& hp = plan_[h];
    if (tm) tm->begin(st, KC_CHOL_FWD, hp.sweep_bytes);
    if (hp.max_nf > 384)
      hipLaunchKernelGGL(forward_kernel<1024>, dim3(hp.nodes.cnt), dim3(1024), (size_t)hp.max_nf * sizeof(double), st,
                         d_nodes_, d_lists_ + hp.nodes.ofs, d_ea_, d_fronts_, d_linv_, d_y_, d_work_);
    else
      hipLaunchKernelGGL(forward_kernel<256>, dim3(hp.nodes.cnt), dim3(256), (size_t)hp.max_nf * sizeof(double), st,
                         d_nodes_, d_lists_ + hp.nodes.ofs, d_ea_, d_fronts_, d_linv_, d_y_, d_work_);
    if (tm) tm->end(st);
  }
  for (int h = nheights_ - 1; h >= 0; --h) {
    const HeightPlan& hp = plan_[h];
    if (tm) tm->begin(st, KC_CHOL_BWD, hp.sweep_bytes);
    if (hp.max_nf > 384)
      hipLaunchKernelGGL(backward_kernel<1024>, dim3(hp.nodes.cnt), dim3(1024), (size_t)hp.max_nf * sizeof(double), st,
                         d_nodes_, d_lists_ + hp.nodes.ofs, d_bdry_, d_fronts_, d_linv_, d_y_);
    else
      hipLaunchKernelGGL(backward_kernel<256>, dim3(hp.nodes.cnt), dim3(256), (size_t)hp.max_nf * sizeof(double), st,
                         d_nodes_, d_lists_ + hp.nodes.ofs, d_bdry_, d_fronts_, d_linv_, d_y_);
    if (tm) tm->end(st);
  }
  hipLaunchKernelGGL(scatter_perm_kernel, dim3(blocks_for(n_)), dim3(TB), 0, st, n_, d_perm_, d_y_, d_x);
  ck(hipGetLastError(), "solve launches");
}

}  // namespace mgb
