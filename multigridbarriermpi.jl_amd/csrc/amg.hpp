// AMG hierarchy + barrier Newton path on one MI355X.
//
// Reference counterparts (all in the absent MultiGridBarrier.jl, evidenced by the reference's tests):
//   amg / AMG            R[l] = blockdiag(subspaces[sv][l]), D[k] = hcat(Z.., op, ..Z)
//                        test/test_d0_construction.jl:82-100
//   barrier f0/f1/f2     test/test_apply_d.jl:44, test/test_column_extract.jl:50-80,
//                        test/test_map_rows_compare.jl:102-123,165-170
//   newton / amgb_step / amgb_core / amgb
//                        SURVEY.md §3.1; solve hook test/test_instrumented_solve.jl:25-99
// Device data layout (HBM, per AMG):
//   Dz, c, v : n x K row-major ;  Y : n x nY row-major ;  z : S*n (column-major n x S, = Julia vec)
//   per level l:  B_l = Dstack*R_l as CSR with rows q*K+k  (Dz = Dz0 + B_l s: ONE SpMV for apply_D)
//                 BT_l = B_l' (gather-form restriction: g = BT_l v, no atomics)
//                 T_l  : (lower-triangle nnz of R_l'HR_l) x (n*nY) plan, A_vals = T_l vec(Y)
#pragma once
#include <memory>
#include <string>
#include <utility>
#include <thread>
#include <exception>
#include <vector>

#include "geometry.hpp"
#include "gpuchol.hpp"
#include "kernels.hpp"
#include "mfchol.hpp"
#include "mg.hpp"

namespace mgb {

void hip_check(hipError_t e, const char* what);

// sum-allreduce of `count` doubles at the device pointer, in place, over the ranks of a row-block sharded
// job.  Called with the context stream idle; must return with the result visible to that stream.
typedef int (*AllreduceFn)(void* user, double* dev_ptr, long long count);

struct Ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  // row-block sharding: replaces the reference's MPI.COMM_WORLD (src:125) + HPCSparseArrays
  // row partition.  world == 1: single GPU, no collective is ever called.
  int rank = 0, world = 1;
  AllreduceFn allreduce = nullptr;
  void* allreduce_user = nullptr;
  long long n_allreduce = 0;
  double allreduce_bytes = 0;
  // library-owned RCCL communicator (comm.cpp): when set, the collectives are ncclAllReduce on `stream` -- no host sync, no callback
  void* rccl_comm = nullptr;
  explicit Ctx(int dev);
  ~Ctx();
  // even_single: run the collective although world == 1 (a one-rank communicator under test)
  void allreduce_sum(double* dev_ptr, long long count, bool even_single = false);
  void set_comm_rccl(const char* id128, int rank, int world);
  void drop_comm();
 private:
  void rccl_allreduce(double* dev_ptr, long long count);
};
void rccl_unique_id(char* out128);

template <class T>
struct DevBuf {
  T* p = nullptr;
  size_t n = 0;
  DevBuf() = default;
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  ~DevBuf() { release(); }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    n = 0;
  }
  void alloc(size_t count) {
    release();
    n = count;
    if (count) hip_check(hipMalloc((void**)&p, count * sizeof(T)), "hipMalloc");
  }
  void upload(const T* h, size_t count) {
    if (count > n) alloc(count);
    if (count) hip_check(hipMemcpy(p, h, count * sizeof(T), hipMemcpyHostToDevice), "H2D");
  }
  void download(T* h, size_t count) const {
    if (count) hip_check(hipMemcpy(h, p, count * sizeof(T), hipMemcpyDeviceToHost), "D2H");
  }
};

template <class T>
struct PinnedBuf {
  T* p = nullptr;
  size_t n = 0;
  PinnedBuf() = default;
  PinnedBuf(const PinnedBuf&) = delete;
  PinnedBuf& operator=(const PinnedBuf&) = delete;
  ~PinnedBuf() {
    if (p) (void)hipHostFree(p);
  }
  void alloc(size_t count) {
    if (p) (void)hipHostFree(p);
    p = nullptr;
    n = count;
    // coherent (uncached on the device side) on purpose: the kernels write results and sequence numbers here with system-scope
    // stores while the host polls -- not left to the runtime's default
    if (count) hip_check(hipHostMalloc((void**)&p, count * sizeof(T), hipHostMallocCoherent), "hipHostMalloc");
  }
};

// Owning device CSR
struct DevCsrOwned {
  DevCsr view;
  DevBuf<int> rowptr, colidx;
  DevBuf<double> vals;
  void upload(const Csr& A);
};

// Owning element-local view of a device CSR (kernels.hpp: DevElCsr); invalid (nel = 0) when the rows do not come in element
// blocks with at most 255 distinct columns each
struct DevElCsrOwned {
  DevElCsr view;
  DevBuf<int> ecols;
  DevBuf<unsigned char> lcol;
  // rows_per_el consecutive rows of A form an element; returns false (and stays invalid) if A does not fit the format
  bool build(const Csr& A, int rows_per_el);
};

// Owning element-operator view of B = D R_l for the matrix-free Hessian product (mg.hpp: DevElOp): structure classes, the dof
// gather lists; B's rowptr / values stay the level's own device CSR.  Invalid when the rows do not come in element blocks.
struct DevElOpOwned {
  DevElOp view;
  DevBuf<int> ecols, cls, dptr, didx;
  DevBuf<unsigned short> c_rowptr, c_tptr;
  DevBuf<unsigned long long> c_ent;
  // block = nodes per element, K = rows of D: rows [e block K, (e + 1) block K) of B form element e
  bool build(const Csr& B, const DevCsr& Bdev, int block, int K, int nY);
  // host copies the element-slab assembly is built from (released by build_assembly)
  std::vector<int> h_ecols, h_cls;
  std::vector<unsigned short> h_tptr;
  std::vector<unsigned long long> h_ent;
};

// Owning element-slab assembly plan (mg.hpp: DevElAsm) on top of an element operator: pair lists per structure class, the gather
// lists of the lower-triangle pattern, the element-matrix scratch.  Replaces the Hessian plan T (46 MB at fem2d L=7, 736 MB at L=9).
struct DevElAsmOwned {
  DevElAsm view;
  DevBuf<int> c_npairs, c_tptr, aptr, aidx;
  DevBuf<unsigned long long> c_terms;
  DevBuf<double> elmat;
  // false (and invalid) if a pair of the elements falls outside the pattern -- the caller then keeps the plan T
  bool build(DevElOpOwned& E, const Csr& Apat, const BarrierParams& P);
};

// Host-side symbolic pieces of the multigrid hierarchy (testable without a GPU)
// P with Rf P = Rc: the level-l basis expressed in the unknowns of level l + 1 (both given on the finest broken nodes); rows of
// P are read off the finest-mesh nodes that carry a single level-(l+1) unknown with weight one (nodal hierarchies)
Csr build_prolongation(const Csr& Rf, const Csr& Rc);
// full symmetric CSR pattern of a lower-triangle pattern; map[k] = index of full entry k in the lower values, diagpos[i] =
// position of (i, i) in the full pattern
Csr sym_full_pattern(const Csr& lower, std::vector<int>& map, std::vector<int>& diagpos);

// V-cycle-preconditioned CG as the Newton linear solver (solver = "pcg")
struct PcgOptions {
  double rtol = 1e-9;      // on sqrt(<r, M r> / <r0, M r0>)
  int maxit = 200;
  int degree = 2;          // operator applications per Chebyshev pre- / post-smoothing
  int power_its = 6;       // power steps per level and Newton matrix for lambda_max(Dinv A), warm-started from the last matrix's vector
  int power_its_cold = 24; // ... and from the fixed start vector (first matrix of a level, or after a breakdown)
  // Chebyshev interval as fractions of the estimate.  The power steps approach lambda_max from below (measured: 0.82-0.95 of it
  // after 6 cold steps, 0.98 after 24, profiles/r3_mg_eig_probe.txt); an interval that ends below lambda_max makes the smoother
  // amplify the top modes and the V-cycle indefinite, hence the margin.
  double lo_frac = 0.12, hi_frac = 1.2;
  int chunk = 4;           // CG iterations enqueued between two looks at the convergence flag
  bool fallback = true;    // direct solve of the step when CG stops without converging
  // after this many consecutive Newton systems on which CG did not converge, the rest of the solve goes to the direct solver
  // without trying (the barrier Hessians only get harder as t grows); 0 = never give up
  int giveup = 3;
  bool assembled_top = false;      // finest level through its assembled matrix instead of the matrix-free product (A/B runs)
};

struct AmgSpec {
  std::vector<std::pair<std::string, std::string>> state_variables;  // (name, subspace key)
  std::vector<std::pair<std::string, std::string>> D;                // (state var, operator key)
};

// Host-only symbolic products of one level (testable without a GPU)
struct LevelPlan {
  int N = 0;
  Csr R;        // (S n) x N
  Csr B;        // (n K) x N, row q*K+k
  Csr BT;       // N x (n K)
  Csr Apat;     // N x N lower-triangle pattern of R'HR (vals unused)
  Csr T;        // nnz(Apat) x (n nY)
  std::vector<double> coords;  // N x dim
  std::vector<unsigned long long> rank_mask;      // sharded plans: dof_rank_masks of the unsharded B (else empty)
};

// Row-block shard [r0, r1) of the node rows (element aligned): local rows of B / R, matching columns of BT / T;
// N, Apat and coords stay global (the Newton unknowns and the factorisation are replicated on every rank).
void shard_rows(int rank, int world, int n, int block, int* r0, int* r1);
// per Newton unknown (column of the UNSHARDED B): bit r set if a row of rank r's row block touches it; empty if world > 64.
// The elimination tree of a sharded job is built along these (MfChol::analyze), so that the matrix entries of a rank's
// subtree are complete on that rank and only separator entries are summed over the ranks.
std::vector<unsigned long long> dof_rank_masks(const Csr& B, int n, int K, int world, int block);
LevelPlan shard_level_plan(const LevelPlan& full, int n, int S, int K, int nY, int r0, int r1);
Csr shard_dstack(const Csr& Dstack, int n, int S, int K, int r0, int r1);

// doubles of reduction scratch an Amg needs: two partials per block of the objective kernels over the local rows, one per
// block of the dots over the (replicated, global) unknowns of the largest level
size_t reduction_scratch_doubles(int n_local, int max_level_unknowns);

Csr build_dstack(const GeometryHost& g, const AmgSpec& spec);
// with_T = false leaves T empty: levels that assemble their Newton matrix element by element (DevElAsmOwned) never need it;
// build_plan_terms fills it in later if a consumer (Float32 evaluation, sharded jobs, host-only tests) asks
LevelPlan build_level_plan(const GeometryHost& g, const AmgSpec& spec, const Csr& Dstack, int level,
                           const BarrierParams& P, bool with_T = true);
void build_plan_terms(LevelPlan& pl, int n, const BarrierParams& P);

// kernel classes timed live with HIP events (KernelTimer) on every 8th Newton step (bracketing every launch
// costs ~14 % of a solve)
enum KernelClass {
  KC_APPLY = 0, KC_F2 = 1, KC_ASSEMBLE = 2, KC_F1 = 3, KC_RESTRICT = 4, KC_F0 = 5,
  KC_CHOL_START = 6, KC_CHOL_STEP = 7, KC_CHOL_BWD_RECT = 8, KC_CHOL_BWD = 9, KC_CHOL_SINGLE = 10, KC_COUNT = 11
};

constexpr double kFracToBoundary = 0.1;   // == oracle FRAC_TO_BOUNDARY
constexpr double kKappaGrowFrac = 0.25;   // == oracle KAPPA_GROW_FRAC
constexpr int kInitialCenteringAttempts = 8;   // == oracle INITIAL_CENTERING_ATTEMPTS
constexpr double kDecrementFrac = 0.01;        // == oracle DECREMENT_FRAC

struct SolveOptions {
  bool host_solve = false;              // true: factor/solve on the host (MfChol), false: on the GPU (GpuChol)
  bool pcg = false;                     // true: V-cycle-preconditioned CG on the GPU (PcgOptions of the Amg)
  bool schedule_all = false;            // false: finest level only; true: coarse -> fine level loop
  bool time_kernels = true;             // bracket kernels with HIP events (a few us of host time per step)
  // end of the t-continuation (both [UPSTREAM-UNVERIFIED], oracle STOP_RULE): false = at the fixed t_stop, the first value of
  // t0 kappa^k beyond 1 / tol, last step clipped (end point independent of the history of kappa reductions); true = the literal
  // loop `while t <= 1 / tol: t <- kappa t`.  Same ts whenever kappa is never reduced.
  bool upstream_stop = false;
  // Newton on the finest level at the intermediate t: true (default) = stagnation of the objective at every t; false = stop once
  // the decrement <g, n> is below kDecrementFrac min w (the path is followed, not resolved) and keep the stagnation rule for the
  // last t, whose centre is the answer (oracle CENTERING; fewer Newton steps at p > 1, more at p = 1: profiles/r3_centering_counts.txt).
  // Phases with an early stop resolve every centre.  [UPSTREAM-UNVERIFIED]
  bool exact_centering = true;
  double tol = 1.4901161193847656e-08;  // sqrt(eps)
  double t0 = 0.1;
  double kappa = 10.0;
  int maxit = 10000;
  int max_newton = 48;
  int verbose = 0;
};

struct SolveStats {
  int L = 0;
  std::vector<long long> its;     // L x nt, column-major (its[l + L*k])
  std::vector<double> ts, c_dot_Dz;
  double t_elapsed = 0, t_setup = 0;
  double time_factor = 0, time_device = 0;
  long long n_factor = 0, n_f0 = 0, n_f1 = 0, n_f2 = 0;
  long long pcg_solves = 0, pcg_iters = 0, pcg_fallbacks = 0;      // solver = pcg: Newton systems, CG iterations, direct fallbacks
  long long pcg_gaveup_at = -1;      // Newton system (1-based count) after which the solve went to the direct solver for good
  double time_pcg = 0;
  // live HIP-event timing of the six kernel classes over the solve (KernelClass order)
  double kern_ms[KC_COUNT] = {};
  double kern_bytes[KC_COUNT] = {};
  long long kern_launches[KC_COUNT] = {};
};

// Event-pair pool: brackets single kernel launches on the context stream; resolved at host syncs.
class KernelTimer {
 public:
  ~KernelTimer();
  void enable(bool on) { on_ = on; sampling_ = true; }
  void sample(bool now) { sampling_ = now; }      // gate: only the Newton steps chosen for sampling are timed
  bool sampling() const { return on_ && sampling_; }
  bool enabled() const { return on_; }
  void begin(hipStream_t st, int cls, double bytes);
  void end(hipStream_t st);
  void collect(SolveStats& st);   // call only after the stream has been synchronised
 private:
  struct Pair {
    hipEvent_t a, b;
    int cls;
    double bytes;
  };
  std::vector<Pair> free_, pending_;
  Pair cur_{};
  bool on_ = false, open_ = false, sampling_ = true;
};

class Amg {
 public:
  Amg(Ctx& ctx, const GeometryHost& g, const AmgSpec& spec, const BarrierParams& P);
  int n() const { return n_; }                 // local rows (= global rows when not sharded)
  int n_global() const { return ng_; }
  int row0() const { return r0_; }
  int S() const { return S_; }
  int K() const { return P_.K; }
  int L() const { return (int)levels_.size(); }
  int level_size(int l) const { return levels_[l]->plan.N; }
  const LevelPlan& plan(int l);   // builds the level on first use
  // build everything a solve needs at level l (-1: every level of the current schedule), incl. the factorisation
  // structures, so that the next solve() is pure compute
  void prepare(int l);
  const BarrierParams& params() const { return P_; }
  // device factorisation of level l: ranks it is split over (1 = replicated), doubles exchanged per solve, launches
  void chol_info(int l, int* split_world, double* exchange_doubles, int* launches);
  bool chol_values_local(int l) { return values_stay_local(level(l)); }
  // feasibility phases: stop the continuation after the first centering at which row `col` of Dz is negative at every node
  // (col < 0: off).  The solve then returns normally with fewer t-steps instead of running to t_stop.
  void set_early_stop(int col) { early_stop_col_ = col; }
  // x-dependent exponent of power-cone term `term` (upstream convex_Euclidian_power with a function p(x)):
  // p_nodes = p at the n_global nodes (>= 1); the barrier kernels then read a = 2 / p and mu(p) per node
  void set_exponents(int term, const double* p_nodes_global);
  // upstream convex_piecewise: mask[q * nterms + c] != 0 iff term c is active at (global) node q; every node keeps >= 1 term
  void set_term_mask(const unsigned char* mask_global);
  // solver = pcg for prepare() and the fine-grained entry points between solves (solve() takes it from its options)
  void set_pcg(bool on) { pcg_ = on; }

  // problem data: c is n x K row-major, z is the S*n vector [u; s]
  void set_c(const double* c_host);
  void set_z(const double* z_host);
  void get_z(double* z_host);

  // fine-grained evaluations at level l, s (N_l host values), barrier parameter t
  //   f0 -> returns objective, also fills parts[2] = {sum w F, sum w c.Dz}
  double f0(int l, const double* s_host, double t, double* parts);
  // line-search trial semantics: objective at s, +inf unless every row keeps >= kFracToBoundary of the cone
  // distance it has at s_ref
  double f0_trial(int l, const double* s_ref_host, const double* s_host, double t);
  void f1(int l, const double* s_host, double t, double* g_host);
  void f2(int l, const double* s_host, double t, double* avals_host);
  // Float32 evaluation of the same three pieces at level l (kernels_f32.hip; float operators and vectors are shadows of the
  // double ones, built on first use).  tpl64 != 0: the double instantiation of the SAME templates instead (tests).
  double f0_f32(int l, const float* s_host, float t);
  void f1_f32(int l, const float* s_host, float t, float* g_host);
  void f2_f32(int l, const float* s_host, float t, float* avals_host);
  void f1_tpl64(int l, const double* s_host, double t, double* g_host);
  void f2_tpl64(int l, const double* s_host, double t, double* avals_host);   // lower-triangle values, plan(l).Apat order
  void apply_D(int l, const double* s_host, double* Dz_host);          // n x K row-major
  // solve (R'HR) nstep = g with the level's multifrontal factorization; returns false if not SPD
  bool solve_host(int l, const double* avals, const double* g, double* nstep);
  // the same on the device (GpuChol): host arrays in/out, for parity tests of the device solver
  bool solve_device(int l, const double* avals, const double* g, double* nstep);

  // full multigrid-barrier solve from the current z; z updated in place
  void solve(const SolveOptions& opt, SolveStats& st);

  // raw pieces for benchmarking: run the device part of one F2 evaluation `reps` times
  struct KernelTimes {
    double apply_ms, f2_ms, assemble_ms, f1_ms, restrict_ms, f0_ms, trial_ms, apply_csr_ms, apply_el;
    double apply_bytes, f2_bytes, assemble_bytes, f1_bytes, restrict_bytes, f0_bytes, trial_bytes;
  };
  KernelTimes time_kernels(int l, int reps, int nrot = 1);      // nrot: rotate over this many distinct copies of every operand

 private:
  struct Level {
    bool built = false, chol_built = false, T_built = false;
    DevElAsmOwned elasm;      // element-slab assembly (single-GPU contexts with element-local operators)
    bool flag_armed = false;      // the pivot flag of gchol is known to be zero (re-armed by the dot kernel of the last solve)
    // captured Newton-step graphs (factorisation chain + <g, n> + both speculative trials), one per set of buffer pointers
    struct StepGraph {
      const void* key[15];
      hipGraphExec_t exec;
    };
    std::vector<StepGraph> step_graphs;
    // the symbolic analysis of the Cholesky factorisation (host only) runs beside the uploads and element tables of the level
    std::thread chol_analysis;
    std::exception_ptr chol_analysis_error;
    bool chol_analyzed = false;
    ~Level() {
      if (chol_analysis.joinable()) chol_analysis.join();
      for (auto& g : step_graphs) (void)hipGraphExecDestroy(g.exec);
    }
    LevelPlan plan;
    DevCsrOwned R, B, BT, T;
    DevElCsrOwned Bel;      // element-local view of B for apply_D on bandwidth-bound meshes
    DevBuf<float> B32, BT32, T32, s32, g32, avals32;      // Float32 shadows (ensure_f32)
    bool f32_built = false;
    MfChol chol;      // symbolic structure (+ host numeric path)
    GpuChol gchol;    // device numeric factorisation / sweeps on the same tree
    DevBuf<double> s, s_trial, s_trial2, s_trial3, g, g_trial, nstep, avals;
    PinnedBuf<double> h_avals, h_g, h_n, h_s;
    // multigrid data of the level (amg_mg.cpp), built on first use
    struct Mg {
      bool elop_tried = false, assembled = false, transfer = false, vectors = false;
      DevElOpOwned elop;              // matrix-free operator B' Y B
      DevBuf<double> elbuf;
      DevCsrOwned A;                  // assembled operator, full symmetric storage (values rewritten per Newton matrix)
      DevBuf<int> amap, diagpos, lo_rowptr, lo_colidx;
      DevCsrOwned P, PT;              // prolongation to level l + 1 (N_{l+1} x N_l) and its transpose
      DevBuf<double> dinv, x, b, r, d0, d1, ev0, ev1, coef, Ainv;
      bool ev_warm = false;           // ev0 holds the dominant vector of an earlier matrix of this level
    };
    std::unique_ptr<Mg> mg;
  };
  struct NewtonResult {
    int k = 0;
    bool converged = false;
  };
  Level& level(int l);            // lazily built
  int level_index(const Level& lv) const;
  void ensure_chol(Level& lv);    // factorisation structures, built on first solve
  void analyze_chol(Level& lv);   // its host-only symbolic part (runs in Level::chol_analysis beside the uploads)
  void ensure_T(Level& lv);       // the Hessian plan T on the device (lazily: only levels without an element-slab assembly, Float32, probes)
  // lower-triangle values of the level's Newton matrix from Y_ into lv.avals: element-slab assembly, or T vec(Y)
  void assemble_values(Level& lv);
  double assemble_bytes(Level& lv);
  void refresh_dz0();
  void dev_apply(Level& lv, const double* s_dev, double* dz);     // dz = Dz0 + B s
  // objective at x = s_dev + alpha * nstep (nstep nullable: x = s_dev); leaves D(z + R x) in dz and x in s_out
  // (nullable) -- one fused launch (trial_f0_kernel)
  double dev_f0(Level& lv, const double* s_dev, double t, double* parts, const double* phi_ref, double* phi_out,
                double* dz, double alpha = 0.0, const double* nstep = nullptr, double* s_out = nullptr);
  double trial_bytes(const Level& lv, bool with_ref) const;
  int fused_trial_rows_ = 64 * 2048;      // trial_f0_kernel on launch-bound meshes (env MGB_FUSED_TRIAL_ROWS)
  void enqueue_f0(Level& lv, const double* s_dev, double alpha, const double* nstep, double* s_out, double* dz,
                  const double* phi_ref, double* phi_out, double* out2, HostSignal sig = HostSignal());      // no host sync
  // gradient from the Dz of the point; returns |g|.  pre != nullptr: also assembles the point's Hessian values behind it
  double dev_f1(Level& lv, const double* dz, double t, double* g_out, SolveStats* st = nullptr, const double** pre = nullptr);
  void enqueue_f2_assemble(Level& lv, const double* dz, SolveStats& st);
  // sharded + device factorisation whose subtrees follow the row partition: Hessian values are NOT summed over the ranks
  bool values_stay_local(Level& lv);
  // ... and inside a solve the Newton vectors (gradient, step, iterate) are OWNER-LOCAL too: valid on this rank's own unknowns and
  // on the replicated top, the only entries its rows touch.  The gradient's collective shrinks to the top entries + one scalar,
  // the step needs none, dots are summed over the owners.
  bool owner_local(Level& lv) { return in_solve_ && values_stay_local(lv); }
  bool in_solve_ = false;
  DevBuf<double> gtop_;
  struct EventHolder {      // owns the event the host waits on for |g|
    hipEvent_t e = nullptr;
    ~EventHolder() {
      if (e) (void)hipEventDestroy(e);
    }
  } ev_f1_holder_;
  hipEvent_t& ev_f1_ = ev_f1_holder_.e;
  // one line-search trial point s - step * nstep with its scratch buffers and (cached) objective value
  struct Trial {
    double* s = nullptr;
    double* phi = nullptr;
    double* dz = nullptr;      // D at the trial point (n x K)
    double step = 0, y = 0;
    bool valid = false;
  };
  // Hessian at s, nstep = H \ g.  With `spec` (device solver only) the first two line-search trials (steps 1 and
  // 1/2, the two points every line search evaluates first) are enqueued behind the solve and read back with the
  // same host synchronisation: two fewer round trips per Newton step.
  bool dev_f2_solve(Level& lv, const double* dz, double t, SolveStats& st, double* inc, Trial* spec = nullptr,
                    const double* pre_assembled = nullptr);
  void enqueue_trial(Level& lv, Trial& T, double step, int slot, HostSignal sig = HostSignal());
  // the first kSpecSteps points every line search may visit (steps 1, 1/2, 1/4), speculated behind the solve: on launch-bound
  // meshes ONE launch of the fused objective kernel evaluates all three (one pass over B); on larger meshes the first two
  // go out as separate bandwidth-shaped launches (a wasted evaluation costs more than a host round trip there)
  int spec_count() const { return n_ <= fused_trial_rows_ ? 3 : 2; }
  void enqueue_spec_trials(Level& lv, Trial* spec, HostSignal sig);
  // single GPU, device solver: factorisation chain, <g, n> (+ pivot flag hand-over) and both speculative trials as ONE
  // hipGraph launch (one per set of buffer pointers; the trial buffers rotate through at most a dozen combinations)
  void launch_step_graph(Level& lv, Trial* spec);
  NewtonResult newton(int l, double t, bool finest, bool final, double lam_tol, int maxit, SolveStats& st, int verbose);
  bool amgb_step(double t, bool final, double lam_tol, int max_newton, std::vector<long long>& its, SolveStats& st, int verbose);
  double c_dot_dz();

  // ---- multigrid / CG (amg_mg.cpp)
 public:
  PcgOptions pcg_opt;
  // H v at the point s of level l: matrix-free (B' (Y o (B v)), element-local) or through the assembled matrix
  void hessian_apply(int l, const double* s_host, const double* v_host, double* out_host, bool matrix_free);
  // `sweeps` Chebyshev-Jacobi smoothing passes of `degree` operator applications each on H(s) x = b at level l from the
  // given x (in / out); lmax > 0: use it as lambda_max(Dinv H), else estimate on the device; returns the value used
  double mg_smooth(int l, const double* s_host, const double* b_host, double* x_host, int degree, int sweeps, double lmax,
                   bool matrix_free);
  void mg_prolong(int l, const double* xc_host, double* xf_host);      // level l -> l + 1
  void mg_restrict(int l, const double* rf_host, double* rc_host);     // level l + 1 -> l
  int mg_coarsest(int top);      // coarsest level of the V-cycle below `top`
  // x = H(s)^{-1} g by V-cycle-preconditioned CG at level l; returns false if it stopped without converging
  bool pcg_solve_linear(int l, const double* s_host, const double* g_host, double* x_host, int* iters, double* relres);
  const Csr& prolongation_host(int l);      // P_l as built on the host (tests)
  // HIP-event timing of the multigrid kernels at level l (Y of the current z), `reps` back-to-back launches rotating over `nrot`
  // distinct copies of the operands (as time_kernels): ms and bytes per call of
  //   [0] H v matrix-free (element pass + dof gather), [1] one Chebyshev step on it (the same two launches with the fused update),
  //   [2] H v through the assembled CSR, [3] prolongation from level l - 1, [4] restriction to level l - 1,
  //   [5] H v as the unfused sequence apply_D-class SpMV on B, block multiply, SpMV on B' would cost it (bytes only; ms = 0)
  // bytes[0..1]: what the kernels move by construction; alg[0..1]: the CSR-based figure of SURVEY.md section 8(d)
  struct MgKernelTimes {
    double ms[6], bytes[6], alg[6];
  };
  MgKernelTimes time_mg_kernels(int l, int reps, int nrot);

 private:
  Level::Mg& mg_of(Level& lv);
  void mg_ensure_vectors(Level& lv);
  bool mg_ensure_elop(Level& lv);
  void mg_ensure_assembled(Level& lv);
  void mg_ensure_transfer(int l);           // P_l, P_l' on level l (to level l + 1)
  void mg_prepare(int top);                 // everything the V-cycle below `top` needs (host + upload)
  bool mg_top_matrix_free(int top);
  void mg_values(int top);                  // numeric setup for the Hessian whose Y is in Y_: operators, diagonals, eigenvalue estimates
  void mg_level_values(Level& lv, bool mf);
  void mg_estimate(Level& lv, bool mf, double lmax_given);
  void mg_apply(Level& lv, bool mf, const MgEpi& epi);
  void mg_smooth_pre(Level& lv, bool mf, const double* b, double* x, double* r, int degree, const double* done);
  void mg_smooth_post(Level& lv, bool mf, const double* b, double* x, double* r, int degree, const double* done);
  void mg_vcycle(int top, int l, const double* b, double* x, const double* done);
  bool pcg_run(Level& lv, int top, const double* g, double* x, SolveStats* st, int* iters, double* relres);
  void eval_Y_at(Level& lv, const double* s_host);
  std::vector<Csr> P_host_;
  DevBuf<double> mg_scal_, mg_scratch_, pcg_r_, pcg_z_, pcg_p_, pcg_Ap_;
  PinnedBuf<double> h_pcg_;
  DevBuf<int> mg_fail_;
  bool mg_inited_ = false;
  bool pcg_ = false;
  double pcg_last_code_ = 0;      // stop code of the last CG: 1 converged, 2 breakdown, 3 iteration cap
  int pcg_bad_streak_ = 0;

  Ctx& ctx_;
  int n_ = 0, S_ = 0;
  int ng_ = 0, r0_ = 0;           // global rows, first local row
  int early_stop_col_ = -1;
  bool slack_negative();
  DevBuf<float> w32_, c32_, Dz0_32_, Dz32_, v32_, Y32_;      // Float32 shadows of the row data
  DevBuf<double> rowF_, rowC_, a_node_, mu_node_;
  DevBuf<unsigned char> term_mask_;
  void ensure_f32(Level& lv);
  BarrierParams P_;
  AmgSpec spec_;
  GeometryHost geo_;              // kept for lazy level construction
  Csr dstack_host_;
  DevCsrOwned Dstack_;
  std::vector<std::unique_ptr<Level>> levels_;
  DevBuf<double> w_, c_, z_, z_save_, Dz0_, Dz0_save_, Dz_, DzA_, DzB_, DzC_, v_, Y_, partials_, scal_, phi_cur_, phi_trial_, phi_trial2_,
      phi_trial3_;
  PinnedBuf<int> h_flag_;
  bool host_solve_ = false;
  PinnedBuf<double> h_scal_;
  double w_min_ = 0;
  bool schedule_all_ = false;
  KernelTimer timer_;
  SolveStats* live_ = nullptr;   // stats object receiving kernel timings during solve()
  void sync_collect(const char* what);
  // Completion signals (single GPU): the last reduction launch of a batch bumps a sequence number in pinned host memory
  // (kernels.hpp: HostSignal); wait_signal() polls it instead of an interrupt-driven stream synchronisation.
  DevBuf<unsigned long long> seq_dev_;
  PinnedBuf<unsigned long long> h_seq_;
  unsigned long long seq_expected_ = 0;
  //  entry points call this first: after an exception between counting a signal and enqueuing its launch the host
  // counter would stay ahead of the device's for ever; one stream synchronisation puts them back in step
  void resync_signals();
  HostSignal next_signal();           // the signal to attach to a launch (null pair on a sharded context); counts it as expected
  void wait_signal(const char* what); // returns when every signalled launch so far has delivered its results to the host
  // pinned-host twin of a slot of scal_ on a single GPU (the reduction kernels write it themselves: no copy launch);
  // null on a sharded context, where the slot is reduced over the ranks first and copied back afterwards
  double* host_scal(double* dev_slot) const { return ctx_.world == 1 ? h_scal_.p + (dev_slot - scal_.p) : nullptr; }
};

}  // namespace mgb
