// Fixed-pattern multifrontal Cholesky for the Newton systems  n = (R' H R) \ g.
//
// Reference behaviour replaced: `MultiGridBarrier.solve(A::HPCSparseMatrix, b::HPCVector) = A \ b`
// -> MUMPS analysis+factor+solve on every Newton step at every level
// (test/test_instrumented_solve.jl:25-28,99; tools/profile_ops.jl:117-126).  BASELINE.json keeps
// the direct solve on the host.  Because the sparsity pattern of R_l' H R_l is fixed per level, the
// symbolic phase (geometric nested dissection from the dof coordinates, elimination tree, front
// index lists, assembly and extend-add maps) runs once per level; each Newton step only scatters
// the new values, runs dense partial factorizations up the tree (OpenMP tasks over subtrees) and
// does the two triangular sweeps.
#pragma once
#include <atomic>
#include <functional>
#include <vector>

#include "sparse.hpp"

namespace mgb {

// Split of the elimination tree over the ranks of a row-block sharded job (the reference factors through MUMPS
// distributed over its MPI ranks, README.md:23, tools/profile_ops.jl:117-126): the `world` subtrees hanging log2(world)
// levels below the root go to one rank each; the separators above them (the "top") are factored redundantly by every
// rank from the subtree roots' Schur complements, which are the only factorisation data that crosses ranks.
struct CholPartition {
  int world = 1;                 // ranks the factorisation is split over (1: every rank factors everything)
  std::vector<int> owner;        // per tree node (postorder): owning rank, -1 = top (replicated)
  std::vector<int> roots;        // the subtree roots, left to right (roots[j] belongs to rank j)
  bool split() const { return world > 1; }
};

class MfChol {
 public:
  // pattern: CSR pattern holding every unordered pair (i,j) exactly once, e.g. the lower triangle
  // (values ignored); coords: N x dim (row-major) dof positions
  // used only to choose the ordering (any values give a correct factorization).
  // rank_mask (nullable, N entries) / world: sharded jobs.  Bit r of rank_mask[i] says that rows of rank r's row block
  // touch unknown i.  The top log2(world) levels of the dissection then follow the ROW PARTITION instead of the geometry:
  // unknowns only ranks of the lower half touch go left, only ranks of the upper half right, the rest is the separator
  // (valid: every matrix entry comes from one element, and an element belongs to one rank).  Subtree r of partition()
  // is then exactly the interior of rank r's row block: every matrix entry of its columns is complete on rank r without
  // any communication, and only entries among separator ("top") unknowns have to be summed over the ranks
  // (rank_aligned()).
  void analyze(const Csr& pattern, const double* coords, int dim, int leaf_size = 64, const unsigned long long* rank_mask = nullptr,
               int world = 1);
  // vals aligned with pattern.colidx of analyze(); returns false on a non-positive pivot.
  bool factor(const double* vals);
  // in-place solve; b has N entries in the ORIGINAL ordering.
  void solve(double* b) const;
  // in-place distributed solve of A x = b over `world` ranks (host mirror of GpuChol's scheme): this rank factors its
  // subtree, the ranks exchange the subtree roots' Schur complements and right-hand-side updates through `allreduce`
  // (sum over ranks, in place, of `count` doubles -- the only collective), every rank factors the top and finishes its
  // own unknowns, and a second allreduce assembles x.  Returns false on a non-positive pivot on ANY rank.
  typedef std::function<void(double*, long long)> Allreduce;
  // vals_local (needs rank_aligned(part.world)): vals holds only THIS rank's row-block contributions; the entries of the
  // top nodes travel with the Schur complements in the first collective and are summed there.
  bool factor_solve_dist(const double* vals, double* b, const CholPartition& part, int rank, const Allreduce& allreduce,
                         bool vals_local = false);
  // true if analyze() was given rank masks for this world and every guided split succeeded
  bool rank_aligned(int world) const { return world > 1 && aligned_world_ == world; }
  // indices into vals of the entries assembled into the replicated top nodes of `part`
  std::vector<int> top_value_indices(const CholPartition& part) const;
  // subtree split for `world` ranks (power of two, complete binary top); world = 1 in the result means "not splittable"
  CholPartition partition(int world) const;
  int size() const { return n_; }
  static int threads();      // worker threads factor() uses (affinity mask, capped at 16; MGB_NUM_THREADS overrides)
  size_t front_doubles() const { return fronts_total_; }
  double factor_flops() const { return flops_; }
  int num_nodes() const { return (int)nodes_.size(); }
  int max_front() const { return max_front_; }
  // elimination tree in postorder: per node own size, front size and parent (-1 for roots)
  void tree(std::vector<int>& ns, std::vector<int>& nf, std::vector<int>& parent) const {
    ns.clear(); nf.clear(); parent.clear();
    for (const Node& nd : nodes_) { ns.push_back(nd.ns); nf.push_back(nd.nf()); parent.push_back(nd.parent); }
  }

 private:
  friend class GpuChol;   // the device factorisation reuses this symbolic structure verbatim
  struct Node {
    int parent = -1;
    int first = 0, ns = 0;      // own dofs: new indices [first, first+ns)
    std::vector<int> bdry;      // boundary dofs (new indices, ascending)
    std::vector<int> children;
    std::vector<int> ea;        // position of bdry[i] inside the parent's front index list
    size_t off = 0;             // offset of the nf x nf column-major front in fronts_
    int nf() const { return ns + (int)bdry.size(); }
  };
  struct Subtree {      // a piece of the elimination tree under construction (node indices local to the piece)
    std::vector<Node> nodes;
    std::vector<std::vector<int>> own;
  };
  int build(std::vector<int>& dofs, int lo, int hi, const Csr& A, const double* coords, int dim, int leaf,
            std::vector<int>& label, std::atomic<int>& next_label, Subtree& sub, int par_depth,
            const unsigned long long* mask = nullptr, int rlo = 0, int rhi = 1);
  void factor_node(int t, const double* vals, bool& ok);
  void forward_node(int t, double* y) const;
  void backward_node(int t, double* y) const;
  int n_ = 0, max_front_ = 0, aligned_world_ = 1, a_nnz_ = 0;
  std::atomic<bool> align_failed_{false};
  double flops_ = 0;
  std::vector<int> perm_, iperm_;       // perm_[new] = old ; iperm_[old] = new
  std::vector<Node> nodes_;             // postorder: children before parents
  std::vector<int> roots_;
  std::vector<std::vector<int>> a_idx_; // per node: indices k into vals ...
  std::vector<std::vector<int>> a_pos_; // ... and their destination inside the front
  std::vector<double> fronts_;          // host numeric fronts, allocated by the first factor()
  size_t fronts_total_ = 0;
};

}  // namespace mgb
