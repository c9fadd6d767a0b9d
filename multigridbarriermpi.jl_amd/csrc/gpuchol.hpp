// Device-resident multifrontal Cholesky for the Newton systems (R'HR) n = g.
//
// Replaces, on the GPU, what the reference does with `A \ b` -> MUMPS on every Newton step
// (test/test_instrumented_solve.jl:25-28,99; tools/profile_ops.jl:117-126).  The sparsity pattern of
// R_l'HR_l is fixed per level, so the host analysis of MfChol (nested-dissection tree, front index lists,
// assembly / extend-add maps) is done once and uploaded; each Newton step then runs, entirely on the
// context stream with no host round trip, ONE dependent chain of launches:
//
//   for each tree height (leaves first):
//     front_start : zero + assemble the front columns, pull both children's Schur complements (fixed
//                   order), factor the first 32x32 pivot block
//     front_step p: (one launch per 32-wide panel) every 64x64 tile of the trailing matrix re-derives the
//                   panel rows it needs (L = A * L11^-T), applies the rank-32 update, and the tile that
//                   owns the next pivot block factors it, so a panel costs one launch
//   for each height (root first): backward sweep  L' x = u
//
// The right-hand side rides along as an extra row of every front (Cholesky of [A b; b' *]), so the
// forward sweep L u = b happens inside the factorisation at no extra launch.  Fronts are stored as
// (nf+1) x (nf+1) column-major squares: the lower triangle holds the working matrix, the upper triangle
// receives the finished rows of L (L[i][k] at row k, column i) -- which is the layout the backward sweep
// wants (unit-stride over k) and makes the panel write race-free.  Everything is gather/pull form:
// no atomics, bitwise reproducible.  Every 32x32 diagonal block of L is kept (row-major, reciprocal
// diagonal) in a side buffer for the substitutions of the panel TRSM and of the backward sweep.
#pragma once
#include <hip/hip_runtime.h>

#include <vector>

#include "mfchol.hpp"

namespace mgb {

class KernelTimer;   // amg.hpp: optional HIP-event bracketing of individual launches
struct Ctx;          // amg.hpp: device, stream and (sharded jobs) the allreduce callback

struct RootXchg {     // one subtree root of a split factorisation: where its Schur complement lives and travels
  long long off;      // front offset
  long long xoff;     // offset of its packed lower triangle (+ right-hand-side row) in the exchange buffer
  int ld, ns, nb;     // leading dimension nf + 1, own columns, boundary size
  int owned;          // 1 on the rank that factors this subtree
};

struct GNode {
  long long off;      // offset of the (nf+1) x (nf+1) column-major front
  long long loff;     // offset of this node's 32x32 pivot blocks of L (one per panel)
  int nf, ns, first, parent;
  int bofs;           // offset of this node's bdry / ea lists (nb entries each)
  int child[2];       // -1 if absent
  int iofs;           // offset of the two (nf+1)-long inverse extend-add maps (-1: leaf)
  int a0, a1;         // range of this node's entries in the assembly list
};

struct StartJob {     // one workgroup of front_start: 32 columns x 256 rows (counted from the chunk's first row) of one front,
                      // with everything of the node and its children inline (no dependent descriptor loads)
  long long off, loff;      // front, first pivot block
  long long boff[2];        // first boundary entry of each child's front (own front offset if absent)
  int cld[2];               // child leading dimensions (0 = no child)
  int nf, ns, iofs, first;
  int node, chunk, rb;
  int a0, a1;         // range of the (column-sorted) assembly list
};

struct StepTile {     // one workgroup of front_step: a 64x64 tile of the trailing matrix of one front
  long long off;      // front offset
  long long loff;     // offset of THIS panel's pivot block
  int nf, ns;
  short ti, tj;
  int pad;            // node index
};

struct SingleTile {   // one workgroup of front_single: a tile of a single-panel front, with everything it needs inline
  long long off, loff;      // front, pivot block
  long long boff[2];        // first boundary entry of each child's front (own front offset if absent)
  int cld[2];               // child leading dimensions (0 = no child)
  int nf, ns, iofs, a0, a1, first;
  short ti, tj;
};

struct RectJob {      // one workgroup of backward_rect: 64 own columns of one front
  int node, chunk;
};

class GpuChol {
 public:
  GpuChol() = default;
  GpuChol(const GpuChol&) = delete;
  GpuChol& operator=(const GpuChol&) = delete;
  ~GpuChol();
  // ctx (nullable): on a sharded context (ctx->world a power of two the tree can be split into) the factorisation is
  // split by subtrees -- this rank factors its subtree, the subtree roots' Schur complements (with their right-hand-side
  // rows) are summed over the ranks (one rank contributes each), every rank factors the replicated top and sweeps back
  // through the top and its own subtree, and a second allreduce assembles x and the pivot flag.  Bitwise the same
  // arithmetic as the unsplit factorisation.
  void build(const MfChol& sym, Ctx* ctx = nullptr);
  bool split() const { return part_.split(); }
  // split AND the tree's top follows the row partition (MfChol::rank_aligned): factor_solve takes THIS rank's row-block
  // contributions to the matrix entries -- the caller does not sum the values over the ranks; the entries of the top
  // nodes ride in the Schur-complement collective and their sums are written back into d_vals.
  bool values_local() const { return vals_local_; }
  double exchange_doubles() const { return (double)xchg_doubles_ + ntop_vals_ + n_ + 1; }
  // d_x = A^{-1} d_b: d_vals = device lower-triangle values in the pattern order given to MfChol::analyze,
  // d_b / d_x device vectors in the ORIGINAL ordering (may alias).
  // flag_armed: the caller guarantees the pivot flag is zero (it re-arms it itself behind the chain): no memset launch
  // d_vals is written only when values_local() (top entries <- their sums over the ranks)
  // values_summed: d_vals already holds the sums over the ranks on every rank (a matrix handed in from outside)
  // x_local (split + values_local only): d_b holds this rank's own-interior entries and the SUMMED entries of the top unknowns
  // (zeros / anything elsewhere); d_x comes back valid on the same set and zero elsewhere, and the pivot flag stays on the device
  // unreduced -- no collective for x: the caller folds the flag into its scalar reduction (Amg: owner-local Newton vectors)
  void factor_solve(hipStream_t st, double* d_vals, const double* d_b, double* d_x, KernelTimer* timer = nullptr,
                    bool flag_armed = false, bool values_summed = false, bool x_local = false);
  // per unknown (original ordering) of a split factorisation: 0 = interior of another rank's subtree, 1 = of this rank's, 2 = top
  const int* unknown_kind() const { return d_kind_orig_; }
  const int* top_unknowns() const { return d_top_unk_; }      // original indices of the top (separator) unknowns
  int ntop_unknowns() const { return ntop_unk_; }
  // the bare launch chain (no flag re-arm, no graph of its own): for callers that capture it into a larger graph together
  // with what follows the solve.  Not for split factorisations (their collectives cannot be captured).
  void enqueue_chain(hipStream_t st, const double* d_vals, const double* d_b, double* d_x);
  int* fail_flag() const { return d_fail_; }   // device int: nonzero after factor_solve() if a pivot was not positive
  int size() const { return n_; }
  double front_bytes() const { return (double)total_front_ * 8; }
  double factor_flops() const { return flops_; }
  int launches_per_solve() const { return launches_; }

 private:
  template <class T>
  T* upload(const std::vector<T>& v);
  void enqueue(hipStream_t st, const double* d_vals, const double* d_b, double* d_x, KernelTimer* timer);
  struct HeightPlan;
  void enqueue_forward(hipStream_t st, const std::vector<HeightPlan>& plan, const double* d_vals, const double* d_b, KernelTimer* tm,
                       int& nprof);
  void enqueue_backward(hipStream_t st, const std::vector<HeightPlan>& plan, double* d_x, KernelTimer* tm, int* nprof = nullptr);
  void factor_solve_split(hipStream_t st, double* d_vals, const double* d_b, double* d_x, KernelTimer* tm, bool values_summed,
                          bool x_local);
  int* d_kind_orig_ = nullptr;
  int* d_top_unk_ = nullptr;
  int ntop_unk_ = 0;
  Ctx* ctx_ = nullptr;
  CholPartition part_;
  long long xchg_doubles_ = 0;
  RootXchg* d_roots_ = nullptr;
  int nroots_ = 0, max_root_nb_ = 0;
  double* d_xchg_ = nullptr;      // Schur exchange buffer (+ ntop_vals_ matrix entries of the top nodes behind it)
  bool vals_local_ = false;
  bool start_pivot_ = true;       // front_start launches carry a dedicated pivot job per front
  int ntop_vals_ = 0;
  int* d_top_idx_ = nullptr;      // indices into d_vals of the entries assembled into the top nodes
  double* d_xsol_ = nullptr;      // n + 1: masked solution + pivot flag
  int* d_own_orig_ = nullptr;     // per unknown (original ordering): 1 if this rank contributes it to the assembled x
  struct GraphEntry {      // captured launch chain for one (values, rhs, solution) pointer triple
    const double* vals;
    const double* b;
    double* x;
    hipGraphExec_t exec;
  };
  std::vector<GraphEntry> graphs_;
  int n_ = 0, nnodes_ = 0, nheights_ = 0, launches_ = 0, max_nf_ = 0;
  long long total_front_ = 0;
  double flops_ = 0;
  // device
  double* d_fronts_ = nullptr;
  double* d_rect_ = nullptr;      // L21' x_bdry of the split backward sweep (n entries, new ordering)
  double* d_linv_ = nullptr;      // 32x32 diagonal (pivot) blocks of L
  double* d_y_ = nullptr;         // solution in the new ordering
  int* d_fail_ = nullptr;
  long long* d_prof_ = nullptr;   // MGB_CHOL_PROF=1: phase stamps of workgroup 0 of every factorisation launch
  [[maybe_unused]] long long* d_wgprof_ = nullptr; // (-DMGB_PROF_PER_WG builds) start / end of every workgroup of every stamped launch
  GNode* d_nodes_ = nullptr;
  int* d_perm_ = nullptr;
  int* d_bdry_ = nullptr;
  int* d_pinv_ = nullptr;        // per parent, per child slot: parent front row -> child boundary row
  int* d_asm_src_ = nullptr;
  int* d_asm_pos_ = nullptr;
  int* d_lists_ = nullptr;        // node lists per height
  GNode* d_hnodes_ = nullptr;     // ... and the node descriptors themselves in that order (leaf / backward launches)
  GNode* d_rnodes_ = nullptr;     // node descriptor per backward_rect job
  StartJob* d_start_ = nullptr;
  StepTile* d_tiles_ = nullptr;
  SingleTile* d_singles_ = nullptr;
  RectJob* d_rectjobs_ = nullptr;
  // host schedule
  struct Range {
    int ofs, cnt;
  };
  struct HeightPlan {
    Range nodes, start, rect;
    std::vector<Range> step;
    Range single_tiles{0, 0};
    std::vector<int> step_npiv;   // leading pivot workgroups of every step launch
    std::vector<int> step_p;      // first panel of the launch
    std::vector<char> step_pair;  // 0: one panel (front_step), 1: two panels fused (front_step2), 2 / 3: two panels as a panel launch and an update launch
    std::vector<double> step_bytes;
    double start_bytes, rect_bytes, tri_bytes;
    int max_nf;
    bool split;      // backward: rectangular part in its own multi-workgroup launch
    bool single;     // every front has one panel: front_single replaces front_start + front_step
    bool narrow = false; // single && at most 8 pivots per front
    bool leaf = false;   // small childless fronts: front_leaf does the whole front in one workgroup
  };
  std::vector<HeightPlan> plan_;       // own nodes (everything when the factorisation is not split)
  std::vector<HeightPlan> plan_top_;   // split only: the replicated separators above the subtrees
  std::vector<void*> allocs_;
};

}  // namespace mgb
