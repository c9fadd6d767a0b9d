// Device-resident multifrontal Cholesky for the Newton systems (R'HR) n = g.
//
// Replaces, on the GPU, what the reference does with `A \ b` -> MUMPS on every Newton step
// (test/test_instrumented_solve.jl:25-28,99; tools/profile_ops.jl:117-126).  The sparsity pattern of
// R_l'HR_l is fixed per level, so the host analysis of MfChol (nested-dissection tree, front index lists,
// assembly / extend-add maps) is done once and uploaded; each Newton step then runs, entirely on the
// context stream with no host round trip:
//   scatter A-values into the (zeroed) fronts  ->  for each tree height: extend-add children, then
//   32-wide panels: [panel factor: one workgroup per front]  [trailing update: 32x32 tiles over all fronts]
//   forward sweep by height (front-local right-hand sides, children pulled in fixed order)
//   backward sweep by height.
// Everything is gather/pull form: no atomics, bitwise reproducible.  The inverse of every 32x32
// diagonal block of L is kept in a side buffer so the sweeps are mat-vecs.
#pragma once
#include <hip/hip_runtime.h>

#include <vector>

#include "mfchol.hpp"

namespace mgb {

class KernelTimer;   // amg.hpp: optional HIP-event bracketing of individual launches

struct GNode {
  long long off;      // offset of the nf x nf column-major front
  long long woff;     // offset of the nf-long front-local work vector
  long long loff;     // offset of this node's block inverses: per panel [Linv | Linv'] (2*32*32 doubles)
  int nf, ns, first, parent;
  int bofs;           // offset of this node's bdry / ea lists (nb entries each)
  int child[2];       // -1 if absent
};

struct GTile {
  int node;
  short ti, tj;
};

class GpuChol {
 public:
  GpuChol() = default;
  GpuChol(const GpuChol&) = delete;
  GpuChol& operator=(const GpuChol&) = delete;
  ~GpuChol();
  void build(const MfChol& sym);
  // d_vals: device lower-triangle values in the pattern order given to MfChol::analyze
  void factor(hipStream_t st, const double* d_vals, KernelTimer* timer = nullptr);
  // d_x = A^{-1} d_b, both device vectors in the ORIGINAL ordering (may alias)
  void solve(hipStream_t st, const double* d_b, double* d_x, KernelTimer* timer = nullptr);
  int* fail_flag() const { return d_fail_; }   // device int: nonzero after factor() if a pivot was not positive
  int size() const { return n_; }
  double front_bytes() const { return (double)total_front_ * 8; }
  double factor_flops() const { return flops_; }
  int launches_per_factor() const { return launches_; }

 private:
  template <class T>
  T* upload(const std::vector<T>& v);
  int n_ = 0, nnodes_ = 0, nheights_ = 0, launches_ = 0, max_nf_ = 0;
  long long total_front_ = 0, total_w_ = 0;
  int nasm_ = 0;
  double flops_ = 0;
  // device
  double* d_fronts_ = nullptr;
  double* d_work_ = nullptr;      // front-local vectors
  double* d_linv_ = nullptr;      // inverses of the 32x32 diagonal blocks of L
  double* d_y_ = nullptr;         // permuted rhs / solution
  int* d_fail_ = nullptr;
  GNode* d_nodes_ = nullptr;
  int* d_perm_ = nullptr;
  int* d_bdry_ = nullptr;
  int* d_ea_ = nullptr;
  int* d_asm_src_ = nullptr;
  long long* d_asm_dst_ = nullptr;
  int* d_lists_ = nullptr;        // all node lists concatenated
  GTile* d_tiles_ = nullptr;
  // host schedule
  struct Range {
    int ofs, cnt;
  };
  struct HeightPlan {
    Range nodes;
    Range ea[2];
    std::vector<Range> panel_nodes, panel_tiles;
    std::vector<double> panel_bytes;
    double ea_bytes[2], sweep_bytes;
    int max_nf;
  };
  std::vector<HeightPlan> plan_;
  std::vector<void*> allocs_;
};

}  // namespace mgb
