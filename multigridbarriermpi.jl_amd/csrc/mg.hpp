// Multigrid pieces of the Newton path for gfx950: the matrix-free Hessian
// product H v = B' (Y o (B v)), Chebyshev-Jacobi smoothing, prolongation / restriction, a dense coarsest-level inverse and the
// device-resident scalars of a V-cycle-preconditioned conjugate-gradient iteration.
//
// The reference has no counterpart of these (its "multigrid" is Newton on nested subspaces with a direct solve per level:
// MultiGridBarrier.solve -> MUMPS, test/test_instrumented_solve.jl:25-28,99); BASELINE.json's north star asks for them, and
// they are graded on the roofline and on end-to-end parity of the solve.  What they must reproduce is the
// operator of the reference's Hessian recipe (test/test_map_rows_compare.jl:102-123,165-170): tests hold H v to the oracle's
// f2 @ v.
#pragma once
#include <hip/hip_runtime.h>

#include "kernels.hpp"

namespace mgb {

// Element operator view of B = D R_l for the matrix-free product: the rows of B come in element blocks (rows_per_el = nodes per
// element x K consecutive rows) that touch at most cmax columns (DevElCsr).  Elements with the same local sparsity share a
// STRUCTURE CLASS: row offsets, 1-byte local column indices and the column-wise (transposed) traversal order live once per
// class in a table that stays cache resident, so per element only its cmax column ids, its class id and its nonzero values
// (B's own CSR values, contiguous per element) come from memory.
struct DevElOp {
  int nel = 0, rows_per_el = 0, cmax = 0, K = 0, block = 0, nnz_max = 0, ncls = 0, N = 0;
  int tpe = 256;                               // threads per element (power of two, 8 .. 256): 256 / tpe elements per workgroup pass
  int slot_doubles = 0;                        // LDS per element, in doubles: xs, vs, ds, us, ys and the element's copy of its class table
  const int* ecols = nullptr;                  // nel x cmax (padded with the element's first column; padded slots are never gathered)
  const int* cls = nullptr;                    // nel
  const int* rowptr = nullptr;                 // B's rowptr (element e's nonzeros start at rowptr[e * rows_per_el])
  const double* vals = nullptr;                // B's values
  const unsigned short* c_rowptr = nullptr;    // ncls x (rows_per_el + 1), relative to the element's first nonzero
  const unsigned short* c_tptr = nullptr;      // ncls x (cmax + 1): column-wise traversal
  // ncls x nnz_max, one 64-bit entry per nonzero: bits 0-15 local nonzero index and 16-31 local row of the p-th entry in
  // column-wise order (sorted by column, then row); bits 32-39 local column of the k-th entry in row-wise order
  const unsigned long long* c_ent = nullptr;
  // gather of the element results: out[i] = sum_{p in dptr[i] .. dptr[i + 1]} elbuf[didx[p]], fixed order -> reproducible
  const int* dptr = nullptr;
  const int* didx = nullptr;
  bool valid() const { return nel > 0; }
};

// Element-slab assembly of the Newton matrix (replaces the plan T of amg.hpp where the rows of B come in element
// blocks): per element the structurally nonzero lower-triangle pairs (i >= j) of its columns -- listed once per structure class
// -- are formed as  sum_q b_qi' Y_q b_qj  into `elmat` (nel x npm), and every lower-triangle entry of A then sums its element
// slots in a fixed order (gather form: reproducible).  Reference recipe: test/test_map_rows_compare.jl:102-123,165-170.
struct DevElAsm {
  int npm = 0, nnzA = 0, slot_doubles = 0, ntm = 0;
  const int* c_npairs = nullptr;               // ncls: pair slots of the class
  // per class and pair slot the products that make up  b_i' Y b_j : offsets c_tptr (ncls x (npm + 1)) into c_terms (ncls x ntm),
  // one 64-bit record per product: bits 0-15 / 16-31 the two local nonzero indices, 32-47 the Y entry (node * nY + slot)
  const int* c_tptr = nullptr;
  const unsigned long long* c_terms = nullptr;
  const int* aptr = nullptr;                   // nnzA + 1
  const int* aidx = nullptr;                   // element slots e * npm + s, elements ascending
  bool valid() const { return npm > 0; }
};
// elmat: nel x npm doubles of scratch; avals: the lower-triangle values in pattern order
void launch_elop_assemble(hipStream_t st, const DevElOp& E, const DevElAsm& A, BarrierParams P, const double* Y, double* elmat,
                          double* avals);

// What happens to t_i = (A vin)_i in the launch that produces it (one pass over the operator per Chebyshev step / CG product):
enum MgMode {
  MG_PLAIN = 0,      // out[i] = t
  MG_FIRST = 1,      // first smoothing step from x = 0: vin = c0 Dinv b (formed on the fly); x = vin; r = b - t; d_new = c1 vin + c2 Dinv r
  MG_STEP = 2,       // vin = d_old: x += d_old; r -= t; d_new = c1 d_old + c2 Dinv r (if has_next); x += d_new too (if add_new)
  MG_RESID = 3,      // vin = x: r = b - t; d_new = c0 Dinv r
  MG_PAP = 4,        // vin = p: out = t; the launch also reduces <p, t> and leaves alpha = rz / <p, Ap> in the CG scalars
  MG_POWER = 5,      // vin = ev * (1 / |ev|): out = Dinv t; the launch reduces |out|^2 -> eigenvalue estimate + next scale
  MG_DIAGINV = 6     // out[i] = 1 / t (the gathered diagonal)
};

// CG / eigenvalue scalars in device memory (doubles): the kernels hand step lengths to each other without the host
enum MgScal {
  SC_RZ = 0, SC_PAP = 1, SC_ALPHA = 2, SC_BETA = 3, SC_RZ0 = 4, SC_TOL2 = 5, SC_ITER = 6, SC_DONE = 7,
  SC_LMAX = 8,       // current estimate of lambda_max(Dinv A) of the level being estimated
  SC_VSCALE = 9,     // 1 / |ev| for the next power step
  SC_MAXIT = 10, SC_COUNT = 16
};
// per level: kChebStride doubles of Chebyshev coefficients: [0] = c0 = 1 / theta, then (c1_k, c2_k) for k = 1 .. degree - 1,
// [15] = the lambda_max they were built from
constexpr int kChebStride = 16;
constexpr int kChebMaxDegree = 7;

struct MgEpi {
  int mode = MG_PLAIN;
  int n = 0;                       // rows
  const double* v = nullptr;       // input vector
  const double* vmul = nullptr;    // optional: vin(j) = vscale * vmul[j] * v[j]
  const double* vscale = nullptr;  // optional device scalar
  const double* coef = nullptr;    // this level's Chebyshev coefficients
  int k = 0;                       // Chebyshev step (coefficient pair index) of MG_STEP / MG_FIRST
  int has_next = 0, add_new = 0;
  const double* dinv = nullptr;
  const double* b = nullptr;
  double* x = nullptr;
  double* r = nullptr;
  double* d_new = nullptr;
  double* out = nullptr;
  double* scal = nullptr;          // MgScal block (MG_PAP, MG_POWER)
  double* scratch = nullptr;       // reduction scratch (kReductionHeader + blocks doubles), MG_PAP / MG_POWER
  const double* done = nullptr;    // &scal[SC_DONE]: nonzero -> the launch returns at once (converged CG keeps its state)
};

// matrix-free level: elbuf = per element B_e' (Y_e o (B_e vin_e)), then the dof gather + epilogue.  elbuf: nel x cmax doubles.
void launch_elop_apply(hipStream_t st, const DevElOp& E, BarrierParams P, const double* Y, double* elbuf, const MgEpi& epi);
// diagonal of B' Y B through the same two launches: out = 1 / diag
void launch_elop_diaginv(hipStream_t st, const DevElOp& E, BarrierParams P, const double* Y, double* elbuf, double* dinv);
// assembled level: the same epilogues behind a CSR product (full symmetric storage)
void launch_csr_apply(hipStream_t st, const DevCsr& A, const MgEpi& epi);
// full[k] = lower[map[k]];  dinv[i] = 1 / full[diagpos[i]]
void launch_expand_sym(hipStream_t st, int nnz_full, const int* map, const double* lower, double* full, int N, const int* diagpos,
                       double* dinv);
// Chebyshev coefficients of one level from the eigenvalue estimate in scal[SC_LMAX]: interval [lo_frac, hi_frac] * lambda
void launch_cheb_coef(hipStream_t st, const double* scal, double* coef, int degree, double lo_frac, double hi_frac);
// start of a power iteration: scal[SC_VSCALE] = 1 / |ev|_D, the norm in which Dinv A is self-adjoint (scratch as the dot kernels)
void launch_power_start(hipStream_t st, int n, const double* ev, const double* dinv, double* scal, double* scratch);
// Ainv = inverse of the SPD matrix given by the lower-triangle CSR (rowptr, colidx, vals) of order N <= kDenseMax, one
// workgroup, Gauss-Jordan in LDS; *fail |= 1 if a pivot is not positive
constexpr int kDenseMax = 128;
void mg_device_init();      // per-device kernel attributes (LDS beyond 64 KiB); call once per device before the first launch
void launch_dense_inverse(hipStream_t st, int N, const int* rowptr, const int* colidx, const double* vals, double* Ainv, int* fail);
// x = Ainv b (N x N row-major, symmetric)
void launch_dense_apply(hipStream_t st, int N, const double* Ainv, const double* b, double* x, const double* done);
// CG pieces (scal: MgScal block; host: pinned mirror {iterations, done, rz, rz0} written by the dot launch, + completion signal)
void launch_pcg_init(hipStream_t st, int n, const double* g, double* x, double* r, double* scal, double rtol, int maxit);
void launch_pcg_update(hipStream_t st, int n, double* x, double* r, const double* p, const double* Ap, const double* scal);
// rz = <r, z>; first call (scal[SC_ITER] == 0): rz0 = rz, beta = 0; else beta = rz / rz_old; convergence -> scal[SC_DONE]
void launch_pcg_dot(hipStream_t st, int n, const double* r, const double* z, double* scal, double* scratch, double* host4,
                    HostSignal sig);
void launch_pcg_p(hipStream_t st, int n, double* p, const double* z, const double* scal);

}  // namespace mgb
