// Native (host) Geometry: the C++ counterpart of MultiGridBarrier's `Geometry{T,Matrix,Vector,
// SparseMatrixCSC}` whose fields the reference converts in native_to_mpi
// (src/MultiGridBarrierMPI.jl:259-338: x, w, operators, subspaces, refine, coarsen).
#pragma once
#include <map>
#include <string>
#include <vector>

#include "sparse.hpp"

namespace mgb {

struct GeometryHost {
  int n = 0;      // broken nodes (rows of x)
  int dim = 0;    // spatial dimension
  int block = 1;  // rows per element (2 in 1-D, 7 in 2-D): rows [e*block,(e+1)*block) share an element
  int L = 0;      // levels
  std::vector<double> x;  // n x dim, row-major
  std::vector<double> w;  // n
  std::map<std::string, Csr> operators;               // "id","dx","dy"
  std::map<std::string, std::vector<Csr>> subspaces;  // "full","dirichlet" -> L matrices, each n x m_l
  std::vector<Csr> refine, coarsen;                   // L each; refine[L-1] = coarsen[L-1] = I
};

// 2^l broken P1 elements on [-1,1] at level l (reference: fem1d, src:547; 16 rows and a 16x7
// Dirichlet subspace at L=3: test/test_nonsquare.jl:28).
GeometryHost fem1d_native(int L);

// Broken P2+bubble triangles (7 nodes/element), red refinement, n = 14*4^(L-1) for the default
// 2-triangle square (docs/src/guide.md:246-253).  K = 3m x 2 row-major vertex list of the coarse
// triangles (docs/src/guide.md:317) or nullptr for the default [-1,1]^2.
GeometryHost fem2d_native(int L, const double* K, int nK_rows);

// Broken Q_k hexahedra ((k+1)^3 nodes per element, k = 1..3) on [-1,1]^3, octree refinement,
// n = 8^(L-1) (k+1)^3 (reference: fem3d, src/MultiGridBarrierMPI.jl:698; k = 3 default src:682-684).
GeometryHost fem3d_native(int L, int k);

}  // namespace mgb
