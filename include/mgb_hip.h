/* mgb_hip.h -- C ABI of libmgb_hip.so, the MI355X-native multigrid-barrier Newton path.
 *
 * Drop-in boundary for the distributed path of sloisel/MultiGridBarrierMPI.jl: each entry point
 * names the reference interface it replaces (file:line relative to the reference repository,
 * src = src/MultiGridBarrierMPI.jl).  All functions return 0 on success or a negative MGB_E_* code;
 * mgb_last_error() returns the message of the last failure on the calling thread.  Nothing throws
 * or aborts across this boundary.  The library owns all device memory behind opaque handles; the
 * caller owns every host buffer.  One host thread drives one context; calls are not re-entrant
 * per context (mirrors "all functions are collective", docs/src/guide.md:63-81).
 *
 * Matrices cross the boundary as CSR with Int32 indices (0-based) and fp64 values -- the layout of
 * the local row block of an HPCSparseMatrix (test/test_dump_matrices.jl:62-71; Ti=Int32 src:260).
 * Dense n x k matrices are row-major unless stated; `z` is the Julia `vec` of the n x S state matrix
 * (column-major, [u; s]).
 */
#ifndef MGB_HIP_H
#define MGB_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MGB_OK 0
#define MGB_E_ARG (-1)      /* bad argument / unknown name / shape mismatch */
#define MGB_E_HIP (-2)      /* HIP runtime failure, or no GPU visible (there is no CPU fallback) */
#define MGB_E_NUMERIC (-3)  /* solver breakdown: non-SPD Hessian, infeasible start, kappa collapse */
#define MGB_E_INTERNAL (-4)

typedef struct mgb_ctx_s* mgb_ctx;   /* device + stream; replaces the HPCBackend instance (src:84-114) */
typedef struct mgb_geo_s* mgb_geo;   /* native Geometry (host); fields of src:318-330 */
typedef struct mgb_amg_s* mgb_amg;   /* AMG hierarchy + barrier problem resident in HBM */
typedef struct mgb_vec_s* mgb_vec;   /* device fp64 vector  == HPCVector.v (src:175) */
typedef struct mgb_csr_s* mgb_csr;   /* device CSR          == HPCSparseMatrix local block (src:216-221) */

const char* mgb_last_error(void);
int mgb_version(void);
/* number of visible HIP devices (0 when there is no GPU); never fails */
int mgb_device_count(void);

/* ---- context ------------------------------------------------------------------------------- */
int mgb_ctx_create(int device_id, mgb_ctx* out);
int mgb_ctx_destroy(mgb_ctx ctx);
int mgb_ctx_synchronize(mgb_ctx ctx);
/* Row-block sharding over `world` ranks (one process + one GPU + one mgb_ctx per rank); replaces the
 * reference's MPI.COMM_WORLD (src:125) + HPCSparseArrays row partition (src:216-221, 259-338).  Every rank
 * uploads the WHOLE geometry; an AMG created on a sharded context keeps the element-aligned row block
 * mgb_shard_rows() assigns to its rank (x, w, z, Dz, c, the barrier kernels and the rows of every operator),
 * replicates the Newton unknowns and the factorisation, and calls `fn` (sum-allreduce of `count` doubles, in
 * place at a device pointer) for the gradient, the Hessian values and the scalar reductions.  The callback is
 * entered with the context stream idle and must return with the result visible to it (e.g. RCCL
 * ncclAllReduce + stream sync, MPI_Allreduce on GPU-aware MPI, torch.distributed.all_reduce).  world == 1
 * (default) never calls it. */
typedef int (*mgb_allreduce_fn)(void* user, double* dev_ptr, long long count);
int mgb_ctx_set_comm(mgb_ctx ctx, int rank, int world, mgb_allreduce_fn fn, void* user);
/* The same sharding with a communicator the LIBRARY owns (RCCL over xGMI; the reference's collectives are in-library too,
 * src:125,132): rank 0 calls mgb_rccl_unique_id (128 bytes) and hands the id to the other ranks by whatever channel the host
 * has (MPI_Bcast, torch.distributed broadcast, a file); every rank then calls mgb_ctx_set_comm_rccl -> ncclCommInitRank.  All
 * collectives of the Newton path become ncclAllReduce calls enqueued on the context stream: no host synchronisation around
 * them and no callback into the host language.  librccl is opened at run time (MGB_RCCL_LIB overrides the name); MGB_E_HIP if
 * it cannot be opened.  world == 1 is allowed (one-rank communicator; mgb_vec_allreduce_sum then runs through RCCL). */
int mgb_rccl_unique_id(char* out128);
int mgb_ctx_set_comm_rccl(mgb_ctx ctx, const char* unique_id128, int rank, int world);
int mgb_ctx_comm_stats(mgb_ctx ctx, long long* calls, double* bytes);
int mgb_shard_rows(int rank, int world, int n, int block, int* r0, int* r1);

/* ---- native geometry (host, setup time) ---------------------------------------------------- *
 * fem1d / fem2d are MultiGridBarrier's geometry builders, called at src:561 and src:628 before
 * native_to_mpi.  K: 3m x 2 row-major coarse triangle vertices or NULL for the default square. */
int mgb_fem1d_native(int L, mgb_geo* out);
int mgb_fem2d_native(int L, const double* K, int nK_rows, mgb_geo* out);
/* fem3d (called at src:698): Q_k hexahedra on the default cube, k = 1..3 (reference default k = 3, src:682-684) */
int mgb_fem3d_native(int L, int k, mgb_geo* out);
/* A Geometry assembled by the caller (native_to_mpi input, src:259-302): create, then add matrices.
 * names: "op:<key>" (n x n), "sub:<key>:<level>" (n x m_l, level 0 = coarsest),
 *        "refine:<level>", "coarsen:<level>".  block = rows per element (1 if unknown). */
int mgb_geo_create(int n, int dim, int L, int block, const double* x, const double* w, mgb_geo* out);
int mgb_geo_set_matrix(mgb_geo g, const char* name, int rows, int cols, const int32_t* rowptr,
                       const int32_t* colidx, const double* vals);
int mgb_geo_destroy(mgb_geo g);
int mgb_geo_dims(mgb_geo g, int* n, int* dim, int* L, int* block);
int mgb_geo_get_xw(mgb_geo g, double* x, double* w);          /* mpi_to_native(geometry), src:355-407 */
int mgb_geo_matrix_info(mgb_geo g, const char* name, int* rows, int* cols, int* nnz);
int mgb_geo_matrix_get(mgb_geo g, const char* name, int32_t* rowptr, int32_t* colidx, double* vals);

/* ---- device vectors / sparse matrices: the array algebra MultiGridBarrier applies - */
int mgb_vec_create(mgb_ctx ctx, int n, const double* host_or_null, mgb_vec* out);  /* HPCVector(v, backend) src:268; amgb_zeros src:116 */
int mgb_vec_free(mgb_vec v);
int mgb_vec_len(mgb_vec v, int* n);
int mgb_vec_upload(mgb_vec v, const double* host);
int mgb_vec_download(mgb_vec v, double* host);                /* Vector(x) gather, src:360; _to_cpu_array src:183-188 */
int mgb_csr_create(mgb_ctx ctx, int rows, int cols, const int32_t* rowptr, const int32_t* colidx,
                   const double* vals, mgb_csr* out);         /* HPCSparseMatrix(S, backend) src:271 */
int mgb_csr_free(mgb_csr A);
int mgb_csr_dims(mgb_csr A, int* rows, int* cols, int* nnz);
int mgb_csr_get(mgb_csr A, int32_t* rowptr, int32_t* colidx, double* vals);   /* SparseMatrixCSC(x) gather, src:371,381 */
/* setup-time sparse algebra MultiGridBarrier applies to M; structural patterns (entries that cancel to 0 are
 * kept: the reference hit a cancellation-dependent sparsity bug, test/test_matrix_addition.jl:21-24).  The symbolic
 * work runs on the host copy of the operands (setup, not the Newton path); the result is device resident. */
int mgb_csr_spgemm(mgb_csr A, mgb_csr B, mgb_csr* out);        /* M*M, M'*M  test/test_basic_ops.jl:39,55; test_nonsquare.jl:83 */
int mgb_csr_transpose(mgb_csr A, mgb_csr* out);                /* materialize_transpose, test/test_transpose_only.jl:38,58 */
int mgb_csr_add(mgb_csr A, double alpha, mgb_csr B, mgb_csr* out);  /* M + alpha*M  test/test_matrix_addition.jl:48-63; scalar*M tools/profile_ops.jl:117 */
int mgb_csr_hcat(int count, const mgb_csr* mats, mgb_csr* out);     /* hcat(M...)  test/test_d0_construction.jl:92-100 */
int mgb_csr_blockdiag(int count, const mgb_csr* mats, mgb_csr* out); /* amgb_blockdiag src:150; test/test_helpers.jl:117-121 */
int mgb_diag(mgb_ctx ctx, mgb_vec z, int m, int n, mgb_csr* out);  /* amgb_diag: spdiagm(m,n,0=>z) src:137-147 */
int mgb_spmv(mgb_csr A, mgb_vec x, mgb_vec y);                /* y = A*x   (HPCSparseMatrix * HPCVector, test_nonsquare.jl:43) */
int mgb_spmv_add(mgb_csr A, mgb_vec x, mgb_vec y0, mgb_vec y); /* y = y0 + A*x */
int mgb_dot(mgb_vec x, mgb_vec y, double* out);               /* dot(w,y) tools/profile_scaling.jl:102 */
int mgb_norm(mgb_vec x, double* out);                         /* norm(x) (2-norm) tools/profile_scaling.jl:89-134 */
int mgb_sum(mgb_vec x, double* out);                          /* sum(x) tools/profile_barrier.jl:45-59,95-114 */
/* out[q] = M[q*K + k] of a row-major n x K device matrix: y[:, j] -> HPCVector, test/test_column_extract.jl:50 */
int mgb_col_extract(mgb_vec M, int n, int K, int k, mgb_vec out);
/* map_rows / map_rows_gpu (src:161-170) for the closures MultiGridBarrier derives from a convex set -- the only closures the
 * Newton path ever maps: which = 0: out[q] = F(Dz_q) (n values, +inf outside the set); 1: out = F1 rows (n x K, row-major);
 * 2: out = F2 rows flattened to K*K columns (n x K*K; column j*K + k, test/test_map_rows_compare.jl:73).  The set is described
 * as in mgb_amg_create_terms; Dz is an n x K row-major device matrix.  Any other closure has no device form: the host side
 * evaluates it on a device -> host copy (the reference's _to_cpu_array trade, src:183-188). */
int mgb_map_rows_barrier(int which, int K, int nterms, const int* kind, const int* nq, const int* idx_q, const int* idx_s,
                         const int* idx_s2, const double* p, const double* coef, const double* off, int n, mgb_vec Dz,
                         mgb_vec out);
int mgb_mul(mgb_vec x, mgb_vec y, mgb_vec out);               /* w .* col, test_column_extract.jl:65 */
int mgb_axpy(mgb_vec x, double alpha, mgb_vec y, mgb_vec out); /* out = x + alpha*y */
int mgb_vec_allreduce_sum(mgb_vec x);                         /* sum over the ranks of a sharded context (no-op for world 1); MPI.Allreduce src:125 */
int mgb_all_isfinite(mgb_vec x, int* out);                    /* amgb_all_isfinite src:121-133 */

/* ---- AMG + barrier problem ------------------------------------------------------------------ *
 * state_vars: S pairs "name\0subspace\0" flattened as 2*S C strings; D: K pairs (state var, operator)
 * (amg(geometry; state_variables, D): layout test/test_d0_construction.jl:82-100).
 * Barrier: power cone {(q,s): s >= |q|^p} on D rows idx_q[0..nq) and idx_s (convex_Euclidian_power).
 * Uploads the geometry to HBM (the native_to_mpi step, src:259-338); levels are built on first use. */
int mgb_amg_create(mgb_ctx ctx, mgb_geo g, int S, const char* const* state_vars, int K, const char* const* D,
                   int nq, const int* idx_q, int idx_s, double p, mgb_amg* out);
/* Barrier of an intersection of 1 or 2 power cones (upstream `intersect` of convex_Euclidian_power sets, as
 * used by parabolic_solve: s1 >= u^2 and s2 >= |grad u|^p).  nq[c], idx_q[3*c + i], idx_s[c], p[c] per cone;
 * idx_s2 (nullable) names an extra D row added to the cone's slack (feasibility phase), -1 for none. */
int mgb_amg_create_cones(mgb_ctx ctx, mgb_geo g, int S, const char* const* state_vars, int K, const char* const* D,
                         int ncones, const int* nq, const int* idx_q, const int* idx_s, const int* idx_s2,
                         const double* p, mgb_amg* out);
/* General barrier menu: an intersection of 1 to 3 convex sets (upstream `intersect`), term c being
 *   kind[c] = 0: the power cone above (nq[c], idx_q[3c + i], idx_s[c], idx_s2[c] (nullable array), p[c]);
 *   kind[c] = 1: the half space  sum_i coef[3c + i] * Dz[:, idx_q[3c + i]] + off[c] > 0,  i < nq[c] <= 3  (upstream
 *                convex_linear with one constant row: bounds and constant obstacles), barrier -log of the affine form;
 *                idx_s[c], p[c] ignored.
 * coef / off may be NULL when every term is a power cone. */
int mgb_amg_create_terms(mgb_ctx ctx, mgb_geo g, int S, const char* const* state_vars, int K, const char* const* D,
                         int nterms, const int* kind, const int* nq, const int* idx_q, const int* idx_s, const int* idx_s2,
                         const double* p, const double* coef, const double* off, mgb_amg* out);
/* x-dependent exponent p(x) of power-cone term `term` (upstream convex_Euclidian_power with a function p):
 * p_nodes[q] = p(x_q) >= 1 at the n (global) nodes; the barrier kernels then use a = 2 / p(x_q) and mu(p(x_q)) per node.
 * Call after mgb_amg_create*, before the first evaluation. */
int mgb_amg_set_exponents(mgb_amg a, int term, const double* p_nodes);
/* upstream convex_piecewise (a convex set that varies in space: at x the intersection of the pieces selected there;
 * [UPSTREAM-UNVERIFIED] semantics): mask[q * nterms + c] != 0 iff barrier term c is active at node q (n
 * global nodes x the terms of mgb_amg_create_terms); an inactive term contributes nothing at that node.  Every node must keep at
 * least one term.  Call after mgb_amg_create*, before the first evaluation; the start must be strictly feasible. */
int mgb_amg_set_term_mask(mgb_amg a, const unsigned char* mask);
int mgb_amg_destroy(mgb_amg a);
int mgb_amg_dims(mgb_amg a, int* n, int* S, int* K, int* L, int* nY);   /* n = LOCAL rows on a sharded context */
int mgb_amg_local_rows(mgb_amg a, int* n_global, int* row0, int* n_local);
/* levels (operators, Hessian plan) are built on first use and the factorisation structures on the first solve;
 * mgb_amg_prepare builds both now for `level` (-1: every level the current schedule visits), so that the next
 * mgb_amg_solve is pure compute */
int mgb_amg_prepare(mgb_amg a, int level);
int mgb_amg_level_size(mgb_amg a, int level, int* N, int* nnz_lower);
/* device factorisation of `level` (built now if needed): *split_world = ranks it is split over by nested-dissection subtrees
 * on a sharded context (1 = replicated: single GPU, or a world the tree cannot be split into), doubles exchanged per Newton
 * system (subtree-root Schur complements + the assembled solution), kernel launches per Newton system */
/* Float32 evaluation of f0 / f1 / f2 at a level (the reference runs Float32 on its Metal backend, test/test_utils.jl:67-88;
 * SURVEY.md section 8 f3): the SpMV and barrier kernels instantiated for float -- float operators, weights, costs and vectors
 * are shadows of the double ones, built on first use -- with s in / g, lower_vals out as float; f0 is summed in double from
 * per-row float terms.  Single-GPU contexts.  There is no Float32 factorisation (fp64 runs at the fp32 vector rate on MI355X
 * and the direct solve is latency-bound): mgb_amg_solve stays in double.  *_template_f64: the double instantiation of the
 * same kernel templates, which must reproduce mgb_amg_f1 / f2 bit for bit (that is how the tests tie the float kernels to
 * the production ones). */
int mgb_amg_f0_f32(mgb_amg a, int level, const float* s, float t, double* f0);
int mgb_amg_f1_f32(mgb_amg a, int level, const float* s, float t, float* g);
int mgb_amg_f2_f32(mgb_amg a, int level, const float* s, float t, float* lower_vals);
int mgb_amg_f1_template_f64(mgb_amg a, int level, const double* s, double t, double* g);
int mgb_amg_f2_template_f64(mgb_amg a, int level, const double* s, double t, double* lower_vals);
int mgb_amg_chol_info(mgb_amg a, int level, int* split_world, double* exchange_doubles, int* launches);
/* *yes = 1 if the subtrees of the split follow the row partition: a rank's Hessian values are then used where they were
 * computed, only the entries among separator unknowns are summed (inside the Schur-complement collective, counted in
 * exchange_doubles), and the allreduce of all nnz values per Newton step is gone (MGB_RANK_ALIGNED=0 restores it) */
int mgb_amg_chol_values_local(mgb_amg a, int level, int* yes);
int mgb_amg_hessian_pattern(mgb_amg a, int level, int32_t* rowptr, int32_t* colidx);  /* lower triangle of R'HR */
int mgb_amg_set_c(mgb_amg a, const double* c);     /* n x K row-major cost (f_grid) */
int mgb_amg_set_z(mgb_amg a, const double* z);     /* S*n, [u; s] */
int mgb_amg_get_z(mgb_amg a, double* z);           /* mpi_to_native(sol).z, src:422-474 */
/* barrier(F).f0/f1/f2 at level `level`, subspace coordinates s (N_l host values), parameter t:
 *   f0: test/test_apply_d.jl:44 + tools/profile_barrier.jl:45-59; parts = {sum w F, sum w c.Dz}
 *   f1: test/test_column_extract.jl:50-80;  f2: test/test_map_rows_compare.jl:102-123,165-170 */
int mgb_amg_apply_D(mgb_amg a, int level, const double* s, double* Dz /* n x K */);
int mgb_amg_f0(mgb_amg a, int level, const double* s, double t, double* y, double* parts2);
/* line-search trial (amgb_all_isfinite semantics, src:121-133, plus the fraction-to-the-boundary rule):
 * y = f0(s) if every row keeps >= 10 % of the cone distance it has at s_ref, else +inf */
int mgb_amg_f0_trial(mgb_amg a, int level, const double* s_ref, const double* s, double t, double* y);
int mgb_amg_f1(mgb_amg a, int level, const double* s, double t, double* g);
int mgb_amg_f2(mgb_amg a, int level, const double* s, double t, double* lower_vals);
/* MultiGridBarrier.solve(A, b) = A \ b (test/test_instrumented_solve.jl:25-28,99), host direct solve */
int mgb_amg_solve_linear(mgb_amg a, int level, const double* lower_vals, const double* g, double* x);
/* the same solve with the device multifrontal Cholesky (csrc/gpuchol.hip): the solver the Newton loop uses
 * by default.  Returns MGB_E_NUMERIC if a pivot is not positive. */
int mgb_amg_solve_linear_gpu(mgb_amg a, int level, const double* lower_vals, const double* g, double* x);
/* Newton linear solver: 0 (default) = GPU multifrontal Cholesky, 1 = host multifrontal Cholesky, 2 = conjugate gradients
 * preconditioned by a V-cycle over the AMG levels, H applied matrix-free (single-GPU contexts; see the multigrid block below) */
int mgb_amg_set_solver(mgb_amg a, int solver);
/* amgb_step level schedule: 0 (default) = Newton on the finest subspace only, 1 = literal coarse -> fine
 * level loop (R_1 ... R_L).  Both end at the same z; see DESIGN.md section 2. */
int mgb_amg_set_schedule(mgb_amg a, int all_levels);
/* end of the t-continuation: 0 (default) = at the fixed t_stop = the first value of t0 kappa^k beyond 1/tol, the last step
 * clipped to land on it (the end point, and z to ~1e-6 at p = 1, then no longer hangs on the history of kappa reductions);
 * 1 = the literal loop of SURVEY.md Appendix A, `while t <= 1/tol: t <- kappa t`.  Neither is confirmed by anything in the
 * reference ([UPSTREAM-UNVERIFIED]; SOL_main.ts is an observable, docs/src/api.md:97-101); both visit the same ts when kappa is
 * never reduced. */
int mgb_amg_set_stop_rule(mgb_amg a, int upstream);
/* Newton's stopping rule on the finest level at the intermediate t: 1 (default) = stagnation of the objective at every t; 0 = stop
 * once the Newton decrement <g, n> is below 0.01 min w -- the path is followed, not resolved -- and keep the stagnation rule for the
 * last t, whose centre is the answer.  Same end point; 15-17 % fewer Newton steps at p = 1.5 / in 3-D, but MORE at p = 1 (fem2d
 * L=7: 466 -> 580), hence the default.  A phase with mgb_amg_set_early_stop resolves every centre.  [UPSTREAM-UNVERIFIED] like
 * every stopping constant (oracle CENTERING). */
int mgb_amg_set_centering(mgb_amg a, int exact);
/* amgb main phase: t-continuation x level loop x Newton; z updated in place */
int mgb_amg_solve(mgb_amg a, double tol, double t0, double kappa, int maxit, int max_newton, int verbose);
/* feasibility phases (SOL_feasibility, src:428-455): make mgb_amg_solve return after the first centering at which row `col`
 * of D z -- the slack of a relaxed problem -- is negative at every node, instead of following the path to t = 1/tol.
 * col = -1 (default): off. */
int mgb_amg_set_early_stop(mgb_amg a, int col);
/* SOL_main fields (docs/src/api.md:97-101) of the last mgb_amg_solve */
int mgb_amg_sol_info(mgb_amg a, int* nt, double* t_elapsed, double* time_factor, long long* counts4);
int mgb_amg_sol_get(mgb_amg a, long long* its /* L x nt col-major */, double* ts, double* c_dot_Dz);
/* live HIP-event timing of the 11 kernel classes accumulated over the last mgb_amg_solve (event pairs
 * recorded on the context stream around single launches, on every 8th Newton step -- bracketing every launch
 * costs ~14 % of a solve): total ms, total algorithmic bytes, launches timed;
 * order = apply_D, barrier_f2, hessian_assemble, barrier_f1, restrict, barrier_f0,
 *         chol_front_start, chol_front_step, chol_backward_rect, chol_backward, chol_front_single */
int mgb_amg_sol_kernels(mgb_amg a, double* ms11, double* bytes11, long long* launches11);
/* per-kernel device timings (HIP events on the context stream around `reps` back-to-back launches, rotating over `nrot`
 * distinct copies of every operand so that a working set of nrot x bytes beyond the 256 MiB Infinity Cache is read from
 * HBM), ms and algorithmic bytes per launch:
 * order = apply_D (as the solve runs it: through the element-local view of B on bandwidth-bound meshes), barrier_f2,
 * hessian_assemble, barrier_f1, restrict, barrier_f0, trial_f0 (the fused trial point + apply_D + barrier_f0 launch of
 * launch-bound meshes), apply_D through the plain CSR kernel; bytes8[7] = 1 if slot 0 used the element-local view */
int mgb_amg_time_kernels(mgb_amg a, int level, int reps, int nrot, double* ms8, double* bytes8);

/* ---- multigrid pieces: SURVEY.md section 8 row a11 / 8(b) `mgb_hessian_apply / mgb_smooth / mgb_prolong / mgb_restrict` ------ *
 * The reference has no smoother: its "multigrid" is Newton on the nested subspaces R_l with MultiGridBarrier.solve -> MUMPS per
 * level (test/test_instrumented_solve.jl:25-28,99; README.md:23).  BASELINE.json's north star asks for prolongation /
 * restriction / smoother on the GPU; they replace nothing one-to-one and serve solver 2 above, which solves the SAME Newton
 * system H n = g (H = R_l' (sum_jk D_j' diag(w y_jk) D_k) R_l, test/test_map_rows_compare.jl:102-123,165-170) iteratively.
 * All take host arrays like mgb_amg_f1 / f2; `s` = the point (N_l subspace coordinates) whose Hessian is meant. */
/* Hv = H(s) v at `level`.  matrix_free != 0: B' (Y o (B v)) element by element -- the element's v and Y staged in LDS, the
 * per-node K x K block applied in registers between the two halves (csrc/mg.hip: elop_apply_kernel); 0: through the assembled
 * matrix (full symmetric CSR).  Both must agree with mgb_amg_f2's matrix times v. */
int mgb_hessian_apply(mgb_amg a, int level, const double* s, const double* v, double* Hv, int matrix_free);
/* `sweeps` Chebyshev-Jacobi passes on H(s) x = b from the given x (in/out), each `degree` (1..7) applications of H; eigenvalue
 * interval [lo, hi] * lambda with lambda = lmax if lmax > 0, else lambda_max(Dinv H) estimated on the device (power steps in the
 * D inner product); *lmax_used (nullable) = the lambda the coefficients were built from. */
int mgb_smooth(mgb_amg a, int level, const double* s, const double* b, double* x, int degree, int sweeps, double lmax,
               int matrix_free, double* lmax_used);
/* transfer between the unknowns of level and level + 1 (R_level = R_{level+1} P): xf = P xc, rc = P' rf */
int mgb_prolong(mgb_amg a, int level, const double* xc, double* xf);
int mgb_restrict(mgb_amg a, int level, const double* rf, double* rc);
/* P of `level` as CSR: sizes first (null arrays), then the arrays */
int mgb_amg_prolongation(mgb_amg a, int level, int* rows, int* cols, int* nnz, int32_t* rowptr, int32_t* colidx, double* vals);
/* x = H(s)^{-1} g by the V-cycle-preconditioned CG the Newton loop runs with solver 2; *converged = 0 if it stopped at maxit */
int mgb_amg_pcg_solve_linear(mgb_amg a, int level, const double* s, const double* g, double* x, int* iters, double* relres,
                             int* converged);
/* CG / V-cycle parameters (a value <= 0, or < 0 for the two flags, keeps the current one): relative tolerance on
 * sqrt(<r, M r>), iteration cap, applications of H per Chebyshev pre-/post-smoothing, power steps per level and Newton matrix,
 * Chebyshev interval fractions, CG iterations enqueued between two looks at the convergence flag, direct solve of a step whose
 * CG did not converge, top level through its assembled matrix instead of the matrix-free product, consecutive non-converged
 * systems after which the rest of the solve goes to the direct solver (0 = keep trying) */
int mgb_amg_set_pcg(mgb_amg a, double rtol, int maxit, int degree, int power_its, double lo_frac, double hi_frac, int chunk,
                    int fallback, int assembled_top, int giveup);
/* of the last mgb_amg_solve with solver 2: counts4 = {Newton systems CG was tried on, CG iterations, direct fallbacks, the
 * Newton system after which the solve went to the direct solver for good (-1: never)}, seconds inside CG */
int mgb_amg_sol_pcg(mgb_amg a, long long* counts4, double* time_s);
/* HIP-event timing of the multigrid kernels at `level` (the Hessian of the current z), `reps` back-to-back launches rotating
 * over `nrot` distinct copies of the operands (as mgb_amg_time_kernels), ms / bytes moved by construction / algorithmic bytes
 * (the CSR-based figure of SURVEY.md section 8d) per call: [0] H v matrix-free, [1] one Chebyshev step on it, [2] H v through the
 * assembled CSR, [3] prolongation from level - 1, [4] restriction to level - 1, [5] bytes of the unfused CSR sequence for H v */
int mgb_amg_time_mg_kernels(mgb_amg a, int level, int reps, int nrot, double* ms6, double* bytes6, double* alg6);
/* coarsest level of the V-cycle whose top is level `top` (the largest level with at most 128 unknowns; dense inverse there) */
int mgb_amg_mg_info(mgb_amg a, int top, int* coarsest);

/* ---- host-only symbolic helpers (no GPU needed; used by the CPU test-suite) ----------------- */

typedef struct mgb_plan_s* mgb_plan;  /* symbolic products of one level: R, B=D*R, B', Hessian plan T */
int mgb_plan_create(mgb_geo g, int S, const char* const* state_vars, int K, const char* const* D, int nq,
                    const int* idx_q, int idx_s, int level, mgb_plan* out);
/* host-only: doubles of reduction scratch an AMG with n_local rows and at most max_level_unknowns (global, replicated)
 * Newton unknowns per level allocates; the CPU suite checks it covers the 8-rank, 3-state-variable configurations */
int mgb_reduction_scratch_doubles(int n_local, int max_level_unknowns, long long* out);
int mgb_plan_destroy(mgb_plan p);
/* host-only: P with R_fine P = R_coarse from two level plans (what mgb_prolong applies); sizes first, then the arrays */
int mgb_plan_prolongation(mgb_plan fine, mgb_plan coarse, int* rows, int* cols, int* nnz, int32_t* rowptr, int32_t* colidx,
                          double* vals);
int mgb_plan_sizes(mgb_plan p, int* N, int* nnz_lower, int* nnz_T, int* nnz_B);
int mgb_plan_pattern(mgb_plan p, int32_t* rowptr, int32_t* colidx);
/* lower_vals = T * vec(Y) evaluated on the host: checks the plan against the reference's Hessian
 * recipe (test/test_matrix_addition.jl:39-95) in the CPU test-suite; not used by the product path */
int mgb_plan_eval_host(mgb_plan p, const double* Y /* n x nY */, double* lower_vals);
/* host-only timing of the multifrontal factorisation/solve of T*vec(Y) on this level's pattern */
/* host-only: the shard of a level plan that rank `rank` of `world` keeps (rows of B / R, columns of BT / T; the
 * pattern stays global), and host products with its B / BT -- the CPU test-suite checks with them that the
 * shards' contributions sum to the unsharded result */
int mgb_plan_shard(mgb_plan p, int S, int K, int rank, int world, int block, mgb_plan* out, int* r0, int* r1);
int mgb_plan_apply_B_host(mgb_plan p, const double* s /* N */, double* Bs /* n_local K */);
int mgb_plan_apply_BT_host(mgb_plan p, const double* v /* n_local K */, double* g /* N */);
int mgb_plan_chol_bench(mgb_plan p, const double* Y, int dim, int reps, double* seconds_per_factor,
                        double* seconds_per_solve, double* flops, double* front_doubles, double* residual);
/* host-only: the product's HOST multifrontal Cholesky (csrc/mfchol.cpp, what solver="host" factors with; worker threads
 * from the affinity mask) on this level's pattern, without a GPU: analyse once, then x = A \ g per Newton matrix.
 * bench.py's cpu_baseline times the Newton path on the host cores with it.  MGB_E_NUMERIC if a pivot is not positive. */
typedef struct mgb_hostchol_s* mgb_hostchol;
int mgb_plan_hostchol_create(mgb_plan p, int dim, mgb_hostchol* out);
int mgb_hostchol_destroy(mgb_hostchol c);
int mgb_hostchol_info(mgb_hostchol c, int* n, int* threads, double* flops);
int mgb_hostchol_factor_solve(mgb_hostchol c, const double* lower_vals, const double* g, double* x);
/* The factorisation split over the ranks of a sharded job (the reference's MUMPS is distributed over its MPI ranks,
 * README.md:23, tools/profile_ops.jl:117-126): the `world` subtrees log2(world) levels below the root of the elimination
 * tree go to one rank each, the separators above them are factored redundantly from the subtree roots' Schur complements.
 * partition: *split_world = ranks really used (1 = tree not splittable, everything replicated), owner[t] = rank of tree
 * node t (postorder, as mgb_plan_chol_tree), -1 = top.  factor_solve_dist: the host mirror of the device scheme; `fn`
 * sum-allreduces HOST doubles in place here.  Every rank passes the same values and right-hand side and gets all of x. */
int mgb_hostchol_partition(mgb_hostchol c, int world, int* split_world, int cap, int* nnodes, int* owner);
int mgb_hostchol_factor_solve_dist(mgb_hostchol c, int rank, int world, mgb_allreduce_fn fn, void* user,
                                   const double* lower_vals, const double* g, double* x);
/* Reduce-to-owner for the matrix entries (row (e); the reference's MUMPS takes distributed entries the same way): analysed
 * with the row partition of a `world`-rank job (`p` = the UNSHARDED plan, K = rows of D, block = rows per element), the top
 * log2(world) levels of the dissection follow the row blocks, subtree r is the interior of rank r's rows and all its matrix
 * entries are complete on rank r.  rank_aligned: *aligned = 1 if that held for every split (else the caller must sum the
 * values over the ranks as before), *top_values = entries among separator unknowns, the only ones that still travel.
 * factor_solve_dist_local: as factor_solve_dist, but `local_lower_vals` holds only THIS rank's row-block contributions
 * (mgb_plan_eval_host of its shard); the top entries ride in the Schur-complement collective.  g is replicated. */
int mgb_plan_hostchol_create_ranked(mgb_plan p, int dim, int K, int world, int block, mgb_hostchol* out);
int mgb_hostchol_rank_aligned(mgb_hostchol c, int world, int* aligned, int* top_values);
int mgb_hostchol_factor_solve_dist_local(mgb_hostchol c, int rank, int world, mgb_allreduce_fn fn, void* user,
                                         const double* local_lower_vals, const double* g, double* x);
/* host-only: the nested-dissection elimination tree of this level's pattern in postorder (children first):
 * own size, front size and parent (-1 = root) of the first min(cap, *nnodes) nodes */
int mgb_plan_chol_tree(mgb_plan p, int dim, int cap, int* nnodes, int* ns, int* nf, int* parent);
int mgb_chol_selftest(int nx, int ny, double* max_residual, double* flops, double* seconds);

#ifdef __cplusplus
}
#endif
#endif /* MGB_HIP_H */
