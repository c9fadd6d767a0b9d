// Host-side CSR container and the handful of symbolic/numeric sparse operations the
// setup phase needs (transpose, SpGEMM, hcat, blockdiag).  Replaces what the reference
// obtains from HPCSparseArrays' `*`, `'`, `hcat`, `blockdiag` on HPCSparseMatrix
// (reference call sites: test/test_d0_construction.jl:82-100, test/test_nonsquare.jl:43-97).
// Indices are Int32, values fp64 (reference default Ti=Int32: src/MultiGridBarrierMPI.jl:260).
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <numeric>
#include <stdexcept>
#include <string>
#include <vector>

#include "errors.hpp"

namespace mgb {

struct Csr {
  int rows = 0, cols = 0;
  std::vector<int> rowptr;   // rows+1
  std::vector<int> colidx;   // nnz, sorted within a row
  std::vector<double> vals;  // nnz
  Csr() : rowptr(1, 0) {}
  Csr(int r, int c) : rows(r), cols(c), rowptr(r + 1, 0) {}
  int nnz() const { return (int)colidx.size(); }
};

struct Triplet {
  int r, c;
  double v;
};

// Build CSR from triplets; duplicates are summed; exact zeros are kept only if keep_zeros.
inline Csr from_triplets(int rows, int cols, std::vector<Triplet> t, bool keep_zeros = false) {
  std::sort(t.begin(), t.end(), [](const Triplet& a, const Triplet& b) { return a.r != b.r ? a.r < b.r : a.c < b.c; });
  Csr A(rows, cols);
  for (size_t i = 0; i < t.size();) {
    size_t j = i;
    double s = 0;
    while (j < t.size() && t[j].r == t[i].r && t[j].c == t[i].c) s += t[j++].v;
    if (s != 0.0 || keep_zeros) {
      A.colidx.push_back(t[i].c);
      A.vals.push_back(s);
      A.rowptr[t[i].r + 1]++;
    }
    i = j;
  }
  for (int r = 0; r < rows; ++r) A.rowptr[r + 1] += A.rowptr[r];
  return A;
}

inline Csr identity(int n) {
  Csr A(n, n);
  A.colidx.resize(n);
  A.vals.assign(n, 1.0);
  std::iota(A.colidx.begin(), A.colidx.end(), 0);
  std::iota(A.rowptr.begin(), A.rowptr.end(), 0);
  return A;
}

inline Csr transpose(const Csr& A) {
  Csr T(A.cols, A.rows);
  T.colidx.resize(A.nnz());
  T.vals.resize(A.nnz());
  for (int k = 0; k < A.nnz(); ++k) T.rowptr[A.colidx[k] + 1]++;
  for (int r = 0; r < T.rows; ++r) T.rowptr[r + 1] += T.rowptr[r];
  std::vector<int> pos(T.rowptr.begin(), T.rowptr.end() - 1);
  for (int r = 0; r < A.rows; ++r)
    for (int k = A.rowptr[r]; k < A.rowptr[r + 1]; ++k) {
      int p = pos[A.colidx[k]]++;
      T.colidx[p] = r;
      T.vals[p] = A.vals[k];
    }
  return T;
}

// C = A * B (Gustavson, sorted rows).  Entries whose value cancels to exactly 0 are dropped
// only when drop_zeros (the reference hit a cancellation-dependent sparsity bug,
// test/test_matrix_addition.jl:21-24: patterns here are structural by default).
inline Csr spgemm(const Csr& A, const Csr& B, bool drop_zeros = false) {
  if (A.cols != B.rows) throw ArgError("spgemm: dimension mismatch");
  Csr C(A.rows, B.cols);
  std::vector<int> mark(B.cols, -1);
  std::vector<double> acc(B.cols, 0.0);
  std::vector<int> list;
  for (int r = 0; r < A.rows; ++r) {
    list.clear();
    for (int k = A.rowptr[r]; k < A.rowptr[r + 1]; ++k) {
      int j = A.colidx[k];
      double a = A.vals[k];
      for (int q = B.rowptr[j]; q < B.rowptr[j + 1]; ++q) {
        int c = B.colidx[q];
        if (mark[c] != r) {
          mark[c] = r;
          acc[c] = 0.0;
          list.push_back(c);
        }
        acc[c] += a * B.vals[q];
      }
    }
    std::sort(list.begin(), list.end());
    for (int c : list)
      if (!drop_zeros || acc[c] != 0.0) {
        C.colidx.push_back(c);
        C.vals.push_back(acc[c]);
      }
    C.rowptr[r + 1] = (int)C.colidx.size();
  }
  return C;
}

// [A B ...] side by side (reference: hcat(Z, D_dx) etc., test_d0_construction.jl:92-100)
inline Csr hcat(const std::vector<const Csr*>& blocks) {
  int rows = blocks[0]->rows, cols = 0;
  for (auto* b : blocks) {
    if (b->rows != rows) throw ArgError("hcat: row mismatch");
    cols += b->cols;
  }
  Csr C(rows, cols);
  for (int r = 0; r < rows; ++r) {
    int off = 0;
    for (auto* b : blocks) {
      for (int k = b->rowptr[r]; k < b->rowptr[r + 1]; ++k) {
        C.colidx.push_back(b->colidx[k] + off);
        C.vals.push_back(b->vals[k]);
      }
      off += b->cols;
    }
    C.rowptr[r + 1] = (int)C.colidx.size();
  }
  return C;
}

// blockdiag(A, B, ...) (reference hook amgb_blockdiag, src:150; test_helpers.jl:117-121)
inline Csr blockdiag(const std::vector<const Csr*>& blocks) {
  int rows = 0, cols = 0;
  for (auto* b : blocks) {
    rows += b->rows;
    cols += b->cols;
  }
  Csr C(rows, cols);
  int ro = 0, co = 0;
  for (auto* b : blocks) {
    for (int r = 0; r < b->rows; ++r) {
      for (int k = b->rowptr[r]; k < b->rowptr[r + 1]; ++k) {
        C.colidx.push_back(b->colidx[k] + co);
        C.vals.push_back(b->vals[k]);
      }
      C.rowptr[ro + r + 1] = (int)C.colidx.size();
    }
    ro += b->rows;
    co += b->cols;
  }
  return C;
}

// C = A + alpha * B on the union pattern (reference: the K^2 - 1 sparse adds of the Hessian recipe,
// test/test_matrix_addition.jl:48-63); structural: entries that cancel to 0 stay in the pattern
inline Csr add(const Csr& A, double alpha, const Csr& B) {
  if (A.rows != B.rows || A.cols != B.cols) throw ArgError("add: shape mismatch");
  Csr C(A.rows, A.cols);
  for (int r = 0; r < A.rows; ++r) {
    int i = A.rowptr[r], j = B.rowptr[r];
    const int ie = A.rowptr[r + 1], je = B.rowptr[r + 1];
    while (i < ie || j < je) {
      const int ca = i < ie ? A.colidx[i] : A.cols, cb = j < je ? B.colidx[j] : A.cols;
      const int c = std::min(ca, cb);
      double v = 0.0;
      if (ca == c) v += A.vals[i++];
      if (cb == c) v += alpha * B.vals[j++];
      C.colidx.push_back(c);
      C.vals.push_back(v);
    }
    C.rowptr[r + 1] = (int)C.colidx.size();
  }
  return C;
}

// rows [r0, r1) of A (row-block shard)
inline Csr row_block(const Csr& A, int r0, int r1) {
  if (r0 < 0 || r1 > A.rows || r0 > r1) throw ArgError("row_block: bad range");
  Csr B(r1 - r0, A.cols);
  const int b = A.rowptr[r0], e = A.rowptr[r1];
  B.colidx.assign(A.colidx.begin() + b, A.colidx.begin() + e);
  B.vals.assign(A.vals.begin() + b, A.vals.begin() + e);
  for (int r = r0; r <= r1; ++r) B.rowptr[r - r0] = A.rowptr[r] - b;
  return B;
}

// rows picked by `pick` (new row i = old row pick[i])
inline Csr row_select(const Csr& A, const std::vector<int>& pick) {
  Csr B((int)pick.size(), A.cols);
  for (size_t i = 0; i < pick.size(); ++i) {
    const int r = pick[i];
    if (r < 0 || r >= A.rows) throw ArgError("row_select: row out of range");
    B.colidx.insert(B.colidx.end(), A.colidx.begin() + A.rowptr[r], A.colidx.begin() + A.rowptr[r + 1]);
    B.vals.insert(B.vals.end(), A.vals.begin() + A.rowptr[r], A.vals.begin() + A.rowptr[r + 1]);
    B.rowptr[i + 1] = (int)B.colidx.size();
  }
  return B;
}

// columns renumbered by map[old] (-1 = not in this shard): entries of dropped columns are removed, or, with
// `strict`, rejected (an operator that couples rows of different shards cannot be row-block sharded).
// The map must be increasing on the kept columns, so rows stay sorted.
inline Csr col_remap(const Csr& A, const std::vector<int>& map, int newcols, bool strict) {
  if ((int)map.size() != A.cols) throw ArgError("col_remap: map size mismatch");
  Csr B(A.rows, newcols);
  for (int r = 0; r < A.rows; ++r) {
    for (int k = A.rowptr[r]; k < A.rowptr[r + 1]; ++k) {
      const int c = map[A.colidx[k]];
      if (c < 0) {
        if (strict) throw ArgError("col_remap: entry couples two shards (operator is not element-local)");
        continue;
      }
      B.colidx.push_back(c);
      B.vals.push_back(A.vals[k]);
    }
    B.rowptr[r + 1] = (int)B.colidx.size();
  }
  return B;
}

inline void spmv_host(const Csr& A, const double* x, double* y) {
  for (int r = 0; r < A.rows; ++r) {
    double s = 0;
    for (int k = A.rowptr[r]; k < A.rowptr[r + 1]; ++k) s += A.vals[k] * x[A.colidx[k]];
    y[r] = s;
  }
}

inline void check_csr(const Csr& A, const char* what) {
  if ((int)A.rowptr.size() != A.rows + 1 || A.rowptr[0] != 0 || A.rowptr[A.rows] != A.nnz() ||
      A.vals.size() != A.colidx.size())
    throw ArgError(std::string("malformed CSR: ") + what);
  for (int r = 0; r < A.rows; ++r) {
    if (A.rowptr[r + 1] < A.rowptr[r]) throw ArgError(std::string("CSR rowptr not monotone: ") + what);
    for (int k = A.rowptr[r]; k < A.rowptr[r + 1]; ++k) {
      if (A.colidx[k] < 0 || A.colidx[k] >= A.cols) throw ArgError(std::string("CSR column out of range: ") + what);
      if (k > A.rowptr[r] && A.colidx[k] <= A.colidx[k - 1]) throw ArgError(std::string("CSR row not sorted/unique: ") + what);
    }
  }
}

}  // namespace mgb
