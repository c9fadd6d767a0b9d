"""multigridbarriermpi.jl_amd -- MI355X-native multigrid-barrier Newton path.

Host-side mirror (Python, because no Julia toolchain exists in the build image) of the operator
surface of sloisel/MultiGridBarrierMPI.jl, on top of the C ABI in ``include/mgb_hip.h``
(``lib/libmgb_hip.so``, hand-written HIP for gfx950).  Names, argument meaning and error behaviour
follow the reference (src = /root/reference/src/MultiGridBarrierMPI.jl):

  fem1d_mpi / fem2d_mpi                 src:559-565, 626-632
  fem1d_mpi_solve / fem2d_mpi_solve     src:594-600, 661-667
  native_to_mpi / mpi_to_native         src:259-338, 355-517
  amgb (re-export)                      src:748-752
  hooks amgb_zeros, amgb_all_isfinite, amgb_diag, amgb_blockdiag, map_rows, map_rows_gpu,
        _raw_array, _to_cpu_array       src:66-192

There is NO CPU fallback: every entry point that computes needs the HIP library and a GPU and
raises otherwise.  (The importable alias of this package is ``mgb_amd``; the directory name
contains a dot and cannot be imported directly.)
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Callable, Dict, List, Optional, Sequence

import numpy as np
import scipy.sparse as sp

from . import _lib
from ._lib import MGBError, call, dptr, f64, i32, iptr

import ctypes as C

__all__ = [
    "fem1d", "fem2d", "fem3d", "fem1d_mpi", "fem2d_mpi", "fem3d_mpi", "fem1d_mpi_solve", "fem2d_mpi_solve",
    "fem3d_mpi_solve", "parabolic_solve", "ParabolicSOL", "native_to_mpi",
    "mpi_to_native", "amgb", "Geometry", "AMGBSOL", "HPCVector", "HPCMatrix", "HPCSparseMatrix",
    "backend_hip", "amgb_zeros", "amgb_all_isfinite", "amgb_diag", "amgb_blockdiag", "map_rows", "map_rows_gpu",
    "_raw_array", "_to_cpu_array", "MGBError", "device_count", "AMG", "amg", "hcat", "BarrierFn", "barrier_functions",
]


def device_count() -> int:
    return int(_lib.load().mgb_device_count())


# --------------------------------------------------------------------------- backend / context


class HPCBackend:
    """One GPU + one stream (replaces HPCBackend{T,Ti,Device,Comm,Solver}, src:84-114)."""

    def __init__(self, device: int = 0):
        h = C.c_void_p()
        call("mgb_ctx_create", int(device), C.byref(h))
        self.handle = h
        self.device = device

    def synchronize(self):
        call("mgb_ctx_synchronize", self.handle)

    rank, world = 0, 1

    def set_comm(self, rank: int, world: int, allreduce):
        """Row-block sharding over `world` ranks (one process + GPU per rank), the
        counterpart of the reference's MPI.COMM_WORLD (src:125).  `allreduce(ptr, count)` must sum-allreduce
        `count` fp64 values in place at the DEVICE pointer `ptr` over all ranks and return only when the
        result is visible to other streams; see `torch_allreduce`.  AMGs created afterwards on this backend
        keep only their rank's rows."""

        def thunk(_user, ptr, count):
            try:
                allreduce(int(ptr), int(count))
                return 0
            except Exception as exc:       # never let an exception cross the C boundary
                import sys
                print("mgb allreduce callback failed: %r" % (exc,), file=sys.stderr)
                return 1

        self._allreduce_cb = _lib.ALLREDUCE_FN(thunk)     # keep the trampoline alive
        call("mgb_ctx_set_comm", self.handle, int(rank), int(world), self._allreduce_cb, None)
        self.rank, self.world = int(rank), int(world)

    def set_comm_rccl(self, rank: int, world: int, unique_id: bytes):
        """The same sharding with a communicator the library owns (mgb_ctx_set_comm_rccl -> ncclCommInitRank): every
        collective of the Newton path is an ncclAllReduce on the context stream, no host synchronisation and no Python in the
        loop.  `unique_id` = the 128 bytes rank 0 got from `rccl_unique_id()`, handed to all ranks by the host's own channel
        (see `rccl_comm_from_torch`)."""
        if len(unique_id) != 128:
            raise ValueError("unique_id must be the 128 bytes of rccl_unique_id()")
        call("mgb_ctx_set_comm_rccl", self.handle, bytes(unique_id), int(rank), int(world))
        self._allreduce_cb = None
        self.rank, self.world = int(rank), int(world)

    def comm_stats(self):
        n, b = C.c_longlong(), C.c_double()
        call("mgb_ctx_comm_stats", self.handle, C.byref(n), C.byref(b))
        return dict(calls=n.value, bytes=b.value)

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                _lib.load().mgb_ctx_destroy(self.handle)
        except Exception:
            pass


def torch_allreduce(dist, device: int, group=None):
    """allreduce callback for HPCBackend.set_comm on top of torch.distributed (plumbing only): backend "nccl"
    (= RCCL over xGMI) reduces in place on the device buffer; any other backend (gloo) stages through the
    host, which is what the single-GPU / CPU rehearsals of the sharded path use."""
    import torch

    class _Raw:
        def __init__(self, ptr, count):
            self.__cuda_array_interface__ = {"shape": (count,), "typestr": "<f8", "data": (ptr, False),
                                             "version": 2, "strides": None}

    on_device = dist.get_backend(group) == "nccl"
    views = {}      # (ptr, count) -> tensor view of the library's buffer: the same few buffers come back every Newton step

    def allreduce(ptr, count):
        t = views.get((ptr, count))
        if t is None:
            if len(views) > 256:
                views.clear()
            t = views[(ptr, count)] = torch.as_tensor(_Raw(ptr, count), device=torch.device("cuda", device))
        if on_device:
            dist.all_reduce(t, group=group)
        else:
            h = t.cpu()
            dist.all_reduce(h, group=group)
            t.copy_(h)
        torch.cuda.synchronize(device)

    return allreduce


def rccl_unique_id() -> bytes:
    """ncclGetUniqueId through the library (call on rank 0, broadcast the 128 bytes)."""
    buf = C.create_string_buffer(128)
    call("mgb_rccl_unique_id", buf)
    return buf.raw


def rccl_comm_from_torch(backend: "HPCBackend", dist, group=None):
    """Give `backend` a library-owned RCCL communicator over the ranks of a torch.distributed job: rank 0 draws the unique id,
    torch.distributed only broadcasts its 128 bytes (setup-time plumbing; nothing of torch stays in the Newton loop)."""
    import torch
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    dev = torch.device("cuda", backend.device) if dist.get_backend(group) == "nccl" else torch.device("cpu")
    t = torch.zeros(129, dtype=torch.uint8, device=dev)      # 128 id bytes + "rank 0 got one": every rank reaches the broadcast
    if rank == 0:
        try:
            t = torch.tensor(list(rccl_unique_id()) + [1], dtype=torch.uint8, device=dev)
        except MGBError:
            pass
    dist.broadcast(t, src=0, group=group)
    raw = bytes(t.cpu().tolist())
    if raw[128] != 1:
        raise MGBError(-2, "rccl_comm_from_torch: rank 0 could not draw an RCCL unique id (librccl not available?)")
    backend.set_comm_rccl(rank, world, raw[:128])


_BACKENDS: Dict[int, HPCBackend] = {}


def backend_hip(device: int = 0) -> HPCBackend:
    """Cached backend instance per device (the reference caches GPU backends the same way, src:84-110)."""
    if device not in _BACKENDS:
        _BACKENDS[device] = HPCBackend(device)
    return _BACKENDS[device]


# --------------------------------------------------------------------------- device array types


class HPCVector:
    """Device fp64 vector (reference HPCVector: `.v` local storage, src:175)."""

    def __init__(self, v, backend: Optional[HPCBackend] = None):
        backend = backend or backend_hip()
        self.backend = backend
        h = C.c_void_p()
        if isinstance(v, (int, np.integer)):
            self.n = int(v)
            call("mgb_vec_create", backend.handle, self.n, None, C.byref(h))
        else:
            a = f64(np.asarray(v).reshape(-1))
            self.n = a.size
            call("mgb_vec_create", backend.handle, self.n, dptr(a), C.byref(h))
        self.handle = h

    def __len__(self):
        return self.n

    @property
    def shape(self):
        return (self.n,)

    def to_numpy(self) -> np.ndarray:
        out = np.empty(self.n)
        call("mgb_vec_download", self.handle, dptr(out))
        return out

    def __array__(self, dtype=None):
        a = self.to_numpy()
        return a if dtype is None else a.astype(dtype)

    def dot(self, other: "HPCVector") -> float:
        out = C.c_double()
        call("mgb_dot", self.handle, other.handle, C.byref(out))
        return out.value

    def norm(self) -> float:            # norm(x), tools/profile_scaling.jl:89-134
        out = C.c_double()
        call("mgb_norm", self.handle, C.byref(out))
        return out.value

    def sum(self) -> float:             # sum(x), tools/profile_barrier.jl:45-59
        out = C.c_double()
        call("mgb_sum", self.handle, C.byref(out))
        return out.value

    def __mul__(self, other):          # w .* y  (test/test_column_extract.jl:65)
        if isinstance(other, HPCVector):
            out = HPCVector(self.n, self.backend)
            call("mgb_mul", self.handle, other.handle, out.handle)
            return out
        return NotImplemented

    def __add__(self, other):
        out = HPCVector(self.n, self.backend)
        call("mgb_axpy", self.handle, 1.0, other.handle, out.handle)
        return out

    def __sub__(self, other):
        out = HPCVector(self.n, self.backend)
        call("mgb_axpy", self.handle, -1.0, other.handle, out.handle)
        return out

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                _lib.load().mgb_vec_free(self.handle)
        except Exception:
            pass


class HPCMatrix:
    """Device dense n x k matrix, row-major (reference HPCMatrix: `.A`, src:176)."""

    def __init__(self, A, backend: Optional[HPCBackend] = None):
        A = f64(np.atleast_2d(np.asarray(A)))
        self.shape = A.shape
        self.backend = backend or backend_hip()
        self._v = HPCVector(A.reshape(-1), self.backend)

    def to_numpy(self) -> np.ndarray:
        return self._v.to_numpy().reshape(self.shape)

    def __array__(self, dtype=None):
        a = self.to_numpy()
        return a if dtype is None else a.astype(dtype)

    def column(self, j: int) -> HPCVector:   # y[:, j] -> HPCVector (test/test_column_extract.jl:50), on the device
        n, K = self.shape
        if not 0 <= j < K:
            raise IndexError("column index out of range")
        out = HPCVector(n, self.backend)
        call("mgb_col_extract", self._v.handle, n, K, int(j), out.handle)
        return out


def hpc_partition(m: int, world: int) -> np.ndarray:
    """1-based offsets of the balanced contiguous split of m rows over `world` ranks (first m mod world ranks hold
    one extra row) -- the shape of the reference's `row_partition` / `.partition` vectors (tools/profile_solve.jl:24).
    HPCSparseArrays' own rule is not vendored in the reference [UPSTREAM-UNVERIFIED]."""
    return np.array([1 + (m // world) * r + min(r, m % world) for r in range(world + 1)], dtype=np.int64)


def hpc_local_block(S, rank: int, world: int, Ti=np.int32) -> dict:
    """The fields rank `rank` of `world` holds for the sparse matrix S in the reference's distributed layout
    (HPCSparseMatrix fields src:216-221; local block as dumped by test/test_dump_matrices.jl:62-71): 1-based
    `row_partition` / `col_partition`, the sorted global ids `col_indices` of the columns the local rows touch, and
    the local rows as the CSC of the transposed block (`colptr` over local rows, `rowval` = compressed column index,
    `nzval`), 1-based like the Julia structs.  Host only; for interop and matrix captures."""
    assert 0 <= rank < world
    S = sp.csr_matrix(S)
    S.sort_indices()
    rpart, cpart = hpc_partition(S.shape[0], world), hpc_partition(S.shape[1], world)
    blk = S[rpart[rank] - 1:rpart[rank + 1] - 1]
    cols = np.unique(blk.indices)
    return dict(row_partition=rpart, col_partition=cpart, col_indices=(cols + 1).astype(Ti),
                colptr=(blk.indptr + 1).astype(Ti), rowval=(np.searchsorted(cols, blk.indices) + 1).astype(Ti),
                nzval=np.array(blk.data, dtype=np.float64), nrows_local=int(blk.shape[0]),
                ncols_compressed=int(cols.size), has_sorted_rows=True)


def hpc_from_local_blocks(blocks) -> sp.csr_matrix:
    """Inverse of `hpc_local_block`: stack the per-rank blocks (in rank order) back into one scipy CSR."""
    rpart, cpart = blocks[0]["row_partition"], blocks[0]["col_partition"]
    rows = []
    for b in blocks:
        ci = np.asarray(b["col_indices"], dtype=np.int64)[np.asarray(b["rowval"], dtype=np.int64) - 1] - 1
        rows.append(sp.csr_matrix((b["nzval"], ci, np.asarray(b["colptr"], dtype=np.int64) - 1),
                                  shape=(b["nrows_local"], int(cpart[-1]) - 1)))
    S = sp.vstack(rows, format="csr")
    assert S.shape[0] == int(rpart[-1]) - 1
    return S


class HPCSparseMatrix:
    """Device CSR matrix (reference HPCSparseMatrix local block, src:216-221).  The library keeps the structure it
    uploaded; `.host` / `to_scipy()` read it back through the C ABI (the reference gathers with
    SparseMatrixCSC(x), src:371).  `A @ B`, `A + B`, `A.T`, `hcat`, `amgb_blockdiag` are the library's setup-time
    sparse algebra (mgb_csr_spgemm / add / transpose / hcat / blockdiag), not scipy."""

    def __init__(self, S, backend: Optional[HPCBackend] = None):
        S = sp.csr_matrix(S, dtype=np.float64)
        S.sort_indices()
        S.sum_duplicates()
        self.shape = S.shape
        self.backend = backend or backend_hip()
        h = C.c_void_p()
        rp, ci, va = i32(S.indptr), i32(S.indices), f64(S.data)
        call("mgb_csr_create", self.backend.handle, S.shape[0], S.shape[1], iptr(rp), iptr(ci), dptr(va), C.byref(h))
        self.handle = h
        self._host = None

    @classmethod
    def _wrap(cls, handle, backend) -> "HPCSparseMatrix":
        out = cls.__new__(cls)
        r, c, nz = C.c_int(), C.c_int(), C.c_int()
        call("mgb_csr_dims", handle, C.byref(r), C.byref(c), C.byref(nz))
        out.shape, out.backend, out.handle, out._host = (r.value, c.value), backend, handle, None
        return out

    @property
    def host(self) -> sp.csr_matrix:
        """scipy copy of the matrix the library holds (fetched once through mgb_csr_get)."""
        if self._host is None:
            r, c, nz = C.c_int(), C.c_int(), C.c_int()
            call("mgb_csr_dims", self.handle, C.byref(r), C.byref(c), C.byref(nz))
            rp = np.empty(r.value + 1, dtype=np.int32)
            ci = np.empty(nz.value, dtype=np.int32)
            va = np.empty(nz.value)
            call("mgb_csr_get", self.handle, iptr(rp), iptr(ci), dptr(va))
            self._host = sp.csr_matrix((va, ci, rp), shape=(r.value, c.value))
        return self._host

    @property
    def nnz(self) -> int:
        nz = C.c_int()
        call("mgb_csr_dims", self.handle, None, None, C.byref(nz))
        return nz.value

    def __matmul__(self, x):
        if isinstance(x, HPCVector):                      # A * x  (test/test_nonsquare.jl:43)
            y = HPCVector(self.shape[0], self.backend)
            call("mgb_spmv", self.handle, x.handle, y.handle)
            return y
        if isinstance(x, HPCSparseMatrix):                # A * B  (test/test_basic_ops.jl:39,55)
            h = C.c_void_p()
            call("mgb_csr_spgemm", self.handle, x.handle, C.byref(h))
            return HPCSparseMatrix._wrap(h, self.backend)
        return NotImplemented

    def __add__(self, other):                             # A + B  (test/test_matrix_addition.jl:48-63)
        if not isinstance(other, HPCSparseMatrix):
            return NotImplemented
        h = C.c_void_p()
        call("mgb_csr_add", self.handle, 1.0, other.handle, C.byref(h))
        return HPCSparseMatrix._wrap(h, self.backend)

    def __sub__(self, other):
        if not isinstance(other, HPCSparseMatrix):
            return NotImplemented
        h = C.c_void_p()
        call("mgb_csr_add", self.handle, -1.0, other.handle, C.byref(h))
        return HPCSparseMatrix._wrap(h, self.backend)

    @property
    def T(self):                                            # lazy Adjoint in the reference; materialised here
        h = C.c_void_p()
        call("mgb_csr_transpose", self.handle, C.byref(h))
        return HPCSparseMatrix._wrap(h, self.backend)

    def to_scipy(self):
        return self.host.copy()

    def local_block(self, rank: int, world: int, Ti=np.int32) -> dict:
        """Rank `rank`'s fields of this matrix in the reference's distributed layout (see `hpc_local_block`)."""
        return hpc_local_block(self.host, rank, world, Ti)

    @staticmethod
    def from_local_blocks(blocks, backend: Optional["HPCBackend"] = None) -> "HPCSparseMatrix":
        return HPCSparseMatrix(hpc_from_local_blocks(blocks), backend)

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                _lib.load().mgb_csr_free(self.handle)
        except Exception:
            pass


def _csr_list_op(name: str, mats) -> "HPCSparseMatrix":
    arr = (C.c_void_p * len(mats))(*[m.handle for m in mats])
    h = C.c_void_p()
    call(name, len(mats), arr, C.byref(h))
    return HPCSparseMatrix._wrap(h, mats[0].backend)


def hcat(*mats: "HPCSparseMatrix") -> "HPCSparseMatrix":
    """hcat(M...) (D0 = hcat(op, Z): test/test_d0_construction.jl:92-100)."""
    return _csr_list_op("mgb_csr_hcat", mats)


class _GeoMatrix(HPCSparseMatrix):
    """A matrix of a geometry the library built itself (mgb_fem*d_native): it already lives in the mgb_geo handle, so
    nothing is copied until somebody asks -- `.host` reads it from the handle, `.handle` (any device operation) uploads it
    on first use.  fem*d_mpi() therefore costs the C++ mesh build only; the solve itself never touches these objects (the
    AMG takes its operators from the mgb_geo handle)."""

    def __init__(self, geo_ref, name: str, backend: "HPCBackend"):
        self._geo_ref, self._name, self.backend, self._host, self._h = geo_ref, name, backend, None, None
        r, c, nz = C.c_int(), C.c_int(), C.c_int()
        call("mgb_geo_matrix_info", geo_ref.handle, name.encode(), C.byref(r), C.byref(c), C.byref(nz))
        self.shape = (r.value, c.value)

    @property
    def host(self) -> sp.csr_matrix:
        if self._host is None:
            self._host = _geo_matrix(self._geo_ref.handle, self._name)
        return self._host

    @property
    def handle(self):
        if self._h is None:
            S = self.host
            h = C.c_void_p()
            rp, ci, va = i32(S.indptr), i32(S.indices), f64(S.data)
            call("mgb_csr_create", self.backend.handle, S.shape[0], S.shape[1], iptr(rp), iptr(ci), dptr(va), C.byref(h))
            self._h = h
        return self._h

    @property
    def nnz(self) -> int:
        return int(self.host.nnz)

    def __del__(self):
        try:
            if self._h is not None:
                _lib.load().mgb_csr_free(self._h)
        except Exception:
            pass


class _GeoRef:
    """Owner of an mgb_geo handle shared by a Geometry and its lazily materialised matrices."""

    def __init__(self, handle):
        self.handle = handle

    def __del__(self):
        try:
            if self.handle is not None:
                _lib.load().mgb_geo_destroy(self.handle)
        except Exception:
            pass


# --------------------------------------------------------------------------- hooks (src:62-192)


def amgb_zeros(like, m, n=None):
    """src:66-75,116: zeros with the storage kind of `like`."""
    if n is None:
        return HPCVector(int(m), getattr(like, "backend", None))
    if isinstance(like, HPCSparseMatrix):
        return HPCSparseMatrix(sp.csr_matrix((m, n)), like.backend)
    return HPCMatrix(np.zeros((m, n)), getattr(like, "backend", None))


def amgb_all_isfinite(z) -> bool:
    """src:121-133: all(isfinite) on the device, one flag back."""
    v = z._v if isinstance(z, HPCMatrix) else z
    out = C.c_int()
    call("mgb_all_isfinite", v.handle, C.byref(out))
    return bool(out.value)


def amgb_diag(like, z, m=None, n=None) -> HPCSparseMatrix:
    """src:137-147: spdiagm(m, n, 0 => z) as a device CSR."""
    backend = getattr(like, "backend", None) or backend_hip()
    if not isinstance(z, HPCVector):
        z = HPCVector(z, backend)
    m = len(z) if m is None else m
    n = len(z) if n is None else n
    h = C.c_void_p()
    call("mgb_diag", backend.handle, z.handle, int(m), int(n), C.byref(h))
    return HPCSparseMatrix._wrap(h, backend)


def amgb_blockdiag(*mats: HPCSparseMatrix) -> HPCSparseMatrix:
    """src:150."""
    return _csr_list_op("mgb_csr_blockdiag", mats)


def _raw_array(x):
    """src:175-176."""
    return x._v if isinstance(x, HPCMatrix) else x


def _to_cpu_array(x):
    """src:183-188: device -> host copy for scalar indexing."""
    return x if isinstance(x, np.ndarray) else x.to_numpy()


def map_rows(f: Callable, A, *args):
    """src:161-163.  Row-wise map over co-partitioned arrays; scalar results -> HPCVector, row results
    -> HPCMatrix.  Arbitrary host closures cannot cross the C ABI: they are evaluated
    on the host on a device->host copy (the same trade as the reference's `_to_cpu_array`, src:183-188)
    and the result is uploaded.  The barrier family used on the Newton hot path never goes through
    here: its F/F1/F2 are the fused HIP kernels behind `AMG.f0/f1/f2`."""
    if isinstance(f, BarrierFn) and len(args) == 1 and isinstance(args[0], HPCMatrix):
        return f.rows(args[0])                      # the barrier family: fused HIP kernels, nothing leaves the device
    arrays = [A, *args]
    backend = next((a.backend for a in arrays if hasattr(a, "backend")), None)
    host = [np.asarray(_to_cpu_array(a), dtype=np.float64) for a in arrays]
    n = host[0].shape[0]
    rows = []
    for i in range(n):
        rows.append(np.asarray(f(*[h[i:i + 1] if h.ndim == 1 else h[i, :] for h in host]), dtype=np.float64))
    if rows and rows[0].ndim == 0:
        return HPCVector(np.array([float(r) for r in rows]), backend)
    return HPCMatrix(np.vstack([r.reshape(1, -1) for r in rows]), backend)


def map_rows_gpu(f: Callable, A, *args):
    """src:168-170."""
    return map_rows(f, A, *args)


# --------------------------------------------------------------------------- Geometry


@dataclass
class Geometry:
    """MultiGridBarrier `Geometry` fields in the reference's order (src:318-330)."""
    discretization: dict
    x: object
    w: object
    subspaces: Dict[str, list]
    operators: Dict[str, object]
    refine: list
    coarsen: list
    _geo: object = field(default=None, repr=False)     # mgb_geo handle (MPI geometries only)
    _geo_ref: object = field(default=None, repr=False) # shared owner of that handle, when the library built the geometry

    def __del__(self):
        try:
            if self._geo is not None and self._geo_ref is None:
                _lib.load().mgb_geo_destroy(self._geo)
        except Exception:
            pass


def _geo_matrix(h, name) -> sp.csr_matrix:
    r, c, nz = C.c_int(), C.c_int(), C.c_int()
    call("mgb_geo_matrix_info", h, name.encode(), C.byref(r), C.byref(c), C.byref(nz))
    rp = np.empty(r.value + 1, dtype=np.int32)
    ci = np.empty(nz.value, dtype=np.int32)
    va = np.empty(nz.value)
    call("mgb_geo_matrix_get", h, name.encode(), iptr(rp), iptr(ci), dptr(va))
    return sp.csr_matrix((va, ci, rp), shape=(r.value, c.value))


def _native_from_handle(h, kind, ops) -> Geometry:
    n, dim, L, block = C.c_int(), C.c_int(), C.c_int(), C.c_int()
    call("mgb_geo_dims", h, C.byref(n), C.byref(dim), C.byref(L), C.byref(block))
    x = np.empty((n.value, dim.value))
    w = np.empty(n.value)
    call("mgb_geo_get_xw", h, dptr(x), dptr(w))
    operators = {k: _geo_matrix(h, "op:" + k) for k in ops}
    subspaces = {k: [_geo_matrix(h, "sub:%s:%d" % (k, l)) for l in range(L.value)] for k in ("dirichlet", "full")}
    refine = [_geo_matrix(h, "refine:%d" % l) for l in range(L.value)]
    coarsen = [_geo_matrix(h, "coarsen:%d" % l) for l in range(L.value)]
    disc = dict(kind=kind, L=L.value, dim=dim.value, block=block.value)
    return Geometry(disc, x, w, subspaces, operators, refine, coarsen)


def fem1d(L: int = 4) -> Geometry:
    """Native 1-D geometry (MultiGridBarrier.fem1d, called at src:561)."""
    h = C.c_void_p()
    call("mgb_fem1d_native", int(L), C.byref(h))
    try:
        return _native_from_handle(h, "fem1d", ("id", "dx"))
    finally:
        call("mgb_geo_destroy", h)


def fem2d(L: int = 2, K=None) -> Geometry:
    """Native 2-D geometry (MultiGridBarrier.fem2d, called at src:628)."""
    h = C.c_void_p()
    if K is None:
        call("mgb_fem2d_native", int(L), None, 0, C.byref(h))
    else:
        Kc = f64(K)
        call("mgb_fem2d_native", int(L), dptr(Kc), int(Kc.shape[0]), C.byref(h))
    try:
        return _native_from_handle(h, "fem2d", ("id", "dx", "dy"))
    finally:
        call("mgb_geo_destroy", h)


def native_to_mpi(g_native: Geometry, Ti=np.int32, backend: Optional[HPCBackend] = None) -> Geometry:
    """src:259-338: convert a native Geometry to device types.  Keys are visited in sorted order as in
    the reference (src:276,286).  Index type is Int32 (src:260); other Ti are rejected."""
    if np.dtype(Ti) != np.dtype(np.int32):
        raise ValueError("native_to_mpi: only Ti=Int32 is supported by the HIP path")
    backend = backend or backend_hip()
    x = f64(np.asarray(g_native.x))
    if x.ndim == 1:
        x = x.reshape(-1, 1)
    w = f64(g_native.w)
    n, dim = x.shape
    L = len(g_native.refine)
    h = C.c_void_p()
    call("mgb_geo_create", n, dim, L, int(g_native.discretization.get("block", 1)), dptr(x), dptr(w), C.byref(h))

    def put(name, S):
        S = sp.csr_matrix(S, dtype=np.float64)
        S.sort_indices()
        S.sum_duplicates()
        rp, ci, va = i32(S.indptr), i32(S.indices), f64(S.data)
        call("mgb_geo_set_matrix", h, name.encode(), S.shape[0], S.shape[1], iptr(rp), iptr(ci), dptr(va))
        return HPCSparseMatrix(S, backend)

    try:
        operators = {k: put("op:" + k, g_native.operators[k]) for k in sorted(g_native.operators)}
        subspaces = {k: [put("sub:%s:%d" % (k, l), S) for l, S in enumerate(g_native.subspaces[k])]
                     for k in sorted(g_native.subspaces)}
        refine = [put("refine:%d" % l, S) for l, S in enumerate(g_native.refine)]
        coarsen = [put("coarsen:%d" % l, S) for l, S in enumerate(g_native.coarsen)]
    except Exception:
        call("mgb_geo_destroy", h)
        raise
    return Geometry(dict(g_native.discretization), HPCMatrix(x, backend), HPCVector(w, backend), subspaces,
                    operators, refine, coarsen, _geo=h)


def _mpi_from_native_handle(h, kind, ops, backend: Optional[HPCBackend], extra=None) -> Geometry:
    """fem*d_mpi without the detour through host matrices: the geometry the library's builder just made IS the uploaded
    geometry (native_to_mpi's job, src:259-338); x and w go to the device now, the matrices when first used (_GeoMatrix)."""
    backend = backend or backend_hip()
    ref = _GeoRef(h)
    n, dim, L, block = C.c_int(), C.c_int(), C.c_int(), C.c_int()
    call("mgb_geo_dims", h, C.byref(n), C.byref(dim), C.byref(L), C.byref(block))
    x = np.empty((n.value, dim.value))
    w = np.empty(n.value)
    call("mgb_geo_get_xw", h, dptr(x), dptr(w))
    lazy = lambda name: _GeoMatrix(ref, name, backend)
    operators = {k: lazy("op:" + k) for k in sorted(ops)}                                    # sorted keys as src:276,286
    subspaces = {k: [lazy("sub:%s:%d" % (k, l)) for l in range(L.value)] for k in ("dirichlet", "full")}
    refine = [lazy("refine:%d" % l) for l in range(L.value)]
    coarsen = [lazy("coarsen:%d" % l) for l in range(L.value)]
    disc = dict(kind=kind, L=L.value, dim=dim.value, block=block.value, **(extra or {}))
    return Geometry(disc, HPCMatrix(x, backend), HPCVector(w, backend), subspaces, operators, refine, coarsen, _geo=h,
                    _geo_ref=ref)


def _check_ti(Ti):
    if np.dtype(Ti) != np.dtype(np.int32):
        raise ValueError("only Ti=Int32 is supported by the HIP path")


def fem1d_mpi(L: int = 4, Ti=np.int32, backend=None) -> Geometry:
    """src:559-565."""
    _check_ti(Ti)
    h = C.c_void_p()
    call("mgb_fem1d_native", int(L), C.byref(h))
    return _mpi_from_native_handle(h, "fem1d", ("id", "dx"), backend)


def fem3d(L: int = 2, k: int = 3) -> Geometry:
    """Native 3-D geometry (MultiGridBarrier.fem3d, called at src:698): Q_k hexahedra, k = 3 default."""
    h = C.c_void_p()
    call("mgb_fem3d_native", int(L), int(k), C.byref(h))
    try:
        g = _native_from_handle(h, "fem3d", ("id", "dx", "dy", "dz"))
        g.discretization["k"] = int(k)
        return g
    finally:
        call("mgb_geo_destroy", h)


def fem3d_mpi(L: int = 2, k: int = 3, Ti=np.int32, backend=None) -> Geometry:
    """src:696-702."""
    _check_ti(Ti)
    h = C.c_void_p()
    call("mgb_fem3d_native", int(L), int(k), C.byref(h))
    return _mpi_from_native_handle(h, "fem3d", ("id", "dx", "dy", "dz"), backend, extra=dict(k=int(k)))


def fem2d_mpi(L: int = 2, K=None, Ti=np.int32, backend=None) -> Geometry:
    """src:626-632."""
    _check_ti(Ti)
    h = C.c_void_p()
    if K is None:
        call("mgb_fem2d_native", int(L), None, 0, C.byref(h))
    else:
        Kc = f64(K)
        call("mgb_fem2d_native", int(L), dptr(Kc), int(Kc.shape[0]), C.byref(h))
    return _mpi_from_native_handle(h, "fem2d", ("id", "dx", "dy"), backend)


# --------------------------------------------------------------------------- AMG + amgb

DEFAULT_STATE = (("u", "dirichlet"), ("s", "full"))
DEFAULT_D = {1: (("u", "id"), ("u", "dx"), ("s", "id")),
             2: (("u", "id"), ("u", "dx"), ("u", "dy"), ("s", "id")),
             3: (("u", "id"), ("u", "dx"), ("u", "dy"), ("u", "dz"), ("s", "id"))}              # src:736
class _RowFn:
    """A per-row function f(x_i) -> vector (what the reference's `f` / `g` kwargs are) that also knows its own
    vectorised form over all rows of x: the default problem data of 57 344 rows is then built in microseconds instead
    of 57 344 Python calls (0.08 s of the 0.45 s setup at fem2d L=7)."""

    def __init__(self, row, grid):
        self._row, self.grid = row, grid

    def __call__(self, x):
        return self._row(x)


def _rows(fn, x) -> np.ndarray:
    """fn evaluated at every row of x -> (n, k) array."""
    if hasattr(fn, "grid"):
        return np.ascontiguousarray(fn.grid(x), dtype=np.float64)
    return np.vstack([np.asarray(fn(xi), dtype=np.float64) for xi in x])


def _const_rows(v):
    v = np.asarray(v, dtype=np.float64)
    return _RowFn(lambda x: v.copy(), lambda x: np.tile(v, (x.shape[0], 1)))


DEFAULT_F = {1: _const_rows([0.5, 0.0, 1.0]), 2: _const_rows([0.5, 0.0, 0.0, 1.0]), 3: _const_rows([0.5, 0.0, 0.0, 0.0, 1.0])}   # src:737
DEFAULT_G = {1: _RowFn(lambda x: np.array([x[0], 2.0]), lambda x: np.column_stack([x[:, 0], np.full(x.shape[0], 2.0)])),
             2: _RowFn(lambda x: np.array([x[0] ** 2 + x[1] ** 2, 100.0]),
                       lambda x: np.column_stack([x[:, 0] ** 2 + x[:, 1] ** 2, np.full(x.shape[0], 100.0)])),
             3: _RowFn(lambda x: np.array([x[0] ** 2 + x[1] ** 2 + x[2] ** 2, 100.0]),
                       lambda x: np.column_stack([x[:, 0] ** 2 + x[:, 1] ** 2 + x[:, 2] ** 2, np.full(x.shape[0], 100.0)]))}   # src:738


def _encode_terms(cones):
    """ctypes arrays describing barrier terms for mgb_amg_create_terms / mgb_map_rows_barrier."""
    kind, nq, iq, isl, is2, pp, coef, off = [], [], [], [], [], [], [], []
    for c in cones:
        if c[0] == "linear":
            _, cidx, ccoef, coff = c
            if not 1 <= len(cidx) <= 3 or len(ccoef) != len(cidx):
                raise ValueError("linear barrier term: 1..3 columns with one coefficient each")
            kind.append(1); nq.append(len(cidx)); iq += (list(cidx) + [0, 0, 0])[:3]; isl.append(0); is2.append(-1)
            pp.append(1.0); coef += (list(map(float, ccoef)) + [0.0, 0.0, 0.0])[:3]; off.append(float(coff))
        else:
            kind.append(0); nq.append(len(c[0]) - 1); iq += (list(c[0][:-1]) + [0, 0, 0])[:3]; isl.append(int(c[0][-1]))
            is2.append(int(c[2]) if len(c) > 2 else -1); pp.append(float(c[1])); coef += [0.0, 0.0, 0.0]; off.append(0.0)
    arr_i = lambda v: (C.c_int * len(v))(*v)
    arr_d = lambda v: (C.c_double * len(v))(*v)
    return (len(cones), arr_i(kind), arr_i(nq), arr_i(iq), arr_i(isl), arr_i(is2), arr_d(pp), arr_d(coef), arr_d(off))


class BarrierFn:
    """One of the three row functions MultiGridBarrier derives from a convex set -- F (which = 0), its gradient F1 (1) or its
    Hessian F2 (2, flattened to K*K columns) -- as an object `map_rows` recognises: `map_rows(fn, x, Dz)` with a device
    matrix Dz then runs the fused HIP kernels (mgb_map_rows_barrier) instead of the host fallback.  This is the closure
    pattern-match of SURVEY.md section 7.2-5: the barrier family has a device form, everything else is evaluated on the host.
    Calling the object on one row (`fn(x_i, dz_i)`, the reference's signature, test/test_apply_d.jl:64,81) works too."""

    def __init__(self, cones, K: int, which: int):
        self.cones, self.K, self.which = list(cones), int(K), int(which)

    def rows(self, Dz: "HPCMatrix"):
        n, K = Dz.shape
        if K != self.K:
            raise ValueError("BarrierFn: Dz has %d columns, the barrier was built for %d" % (K, self.K))
        width = (1, K, K * K)[self.which]
        out = HPCVector(n * width, Dz.backend)
        call("mgb_map_rows_barrier", self.which, K, *_encode_terms(self.cones), n, Dz._v.handle, out.handle)
        if self.which == 0:
            return out
        M_ = HPCMatrix.__new__(HPCMatrix)
        M_.shape, M_.backend, M_._v = (n, width), Dz.backend, out
        return M_

    def __call__(self, x_row, dz_row):
        r = self.rows(HPCMatrix(np.asarray(dz_row, dtype=np.float64).reshape(1, -1))).to_numpy()
        return float(r[0]) if self.which == 0 else r[0]


def barrier_functions(cones, K: int):
    """(F, F1, F2) of the intersection of the given terms (see `AMG` for the term syntax), acting on rows of an n x K Dz."""
    return tuple(BarrierFn(cones, K, w) for w in (0, 1, 2))


class AMG:
    """AMG hierarchy + barrier problem resident in HBM (upstream `amg` + `barrier`)."""

    def __init__(self, geometry: Geometry, state_variables=DEFAULT_STATE, D=None, p: float = 1.0, idx=None,
                 cones=None, select=None):
        """`cones` = the terms of the barrier (an intersection of up to three convex sets, upstream `intersect`):
        (idx, p) | (idx, p, idx_s2) -- the power cone s >= |q|^p on the D rows idx = (q_1..q_d, s) (convex_Euclidian_power);
        ("linear", idx, coef, off)  -- the half space sum_i coef[i] * Dz[:, idx[i]] + off > 0 (convex_linear with one constant
        row: bounds, constant obstacles).  Default: one power cone on the last dim+1 rows of D with exponent p."""
        if geometry._geo is None:
            raise TypeError("AMG needs an MPI geometry (use native_to_mpi / fem*d_mpi)")
        dim = geometry.discretization["dim"]
        D = DEFAULT_D[dim] if D is None else D
        K = len(D)
        if cones is None:
            if idx is None:
                idx = list(range(K - dim - 1, K))       # convex_Euclidian_power(idx=2:dim+2)
            cones = [(list(idx), float(p) if np.isscalar(p) else p)]      # p may be a function p(x) or per-node values
        # x-dependent exponents (upstream convex_Euclidian_power with a function p(x)): a callable or an array of per-node values in
        # a power-cone term is evaluated at the nodes and handed over with mgb_amg_set_exponents after the AMG exists
        self.geometry = geometry
        node_p, cones = {}, list(cones)
        for ti, c in enumerate(cones):
            if c[0] != "linear" and not np.isscalar(c[1]):
                pv = c[1]
                pn = (np.array([float(pv(xi)) for xi in geometry.x.to_numpy()]) if callable(pv) else f64(pv).reshape(-1))
                if pn.size != geometry.x.shape[0] or not np.all(pn >= 1.0):
                    raise ValueError("p(x) must give one value >= 1 per node")
                node_p[ti] = pn
                cones[ti] = (c[0], float(pn[0])) + tuple(c[2:])
        p = float(p) if np.isscalar(p) else (float(node_p[0][0]) if 0 in node_p else 1.0)
        self.p_nodes = node_p.get(0)
        self.state_variables, self.D, self.p, self.cones = tuple(state_variables), tuple(D), float(p), list(cones)
        power = [c for c in cones if c[0] != "linear"]
        self.idx = list(power[0][0]) if power else []
        backend = geometry.x.backend
        h = C.c_void_p()
        call("mgb_amg_create_terms", backend.handle, geometry._geo, len(state_variables), _lib.str_array(state_variables), K,
             _lib.str_array(D), *_encode_terms(cones), C.byref(h))
        self.handle = h
        n, S, K_, L, nY = (C.c_int() for _ in range(5))
        call("mgb_amg_dims", h, C.byref(n), C.byref(S), C.byref(K_), C.byref(L), C.byref(nY))
        self.S, self.K, self.L, self.nY = S.value, K_.value, L.value, nY.value
        ng, r0, nl = C.c_int(), C.c_int(), C.c_int()
        call("mgb_amg_local_rows", h, C.byref(ng), C.byref(r0), C.byref(nl))
        # n = global rows (what set_c / set_z / get_z exchange on every rank); a sharded AMG (backend.set_comm)
        # evaluates apply_D on its own rows [row0, row0 + n_local) only
        self.n, self.row0, self.n_local = ng.value, r0.value, nl.value
        for ti, pn in node_p.items():
            call("mgb_amg_set_exponents", h, int(ti), dptr(pn))
        if select is not None:      # upstream convex_piecewise: term c is active at x iff select(x)[c]
            mask = np.ascontiguousarray([[1 if b else 0 for b in select(xi)] for xi in geometry.x.to_numpy()], dtype=np.uint8)
            if mask.shape != (self.n, len(cones)):
                raise ValueError("select(x) must give one flag per barrier term")
            call("mgb_amg_set_term_mask", h, mask.ctypes.data_as(C.POINTER(C.c_ubyte)))

    def prepare(self, l=-1):
        """Build the level(s) and the factorisation structures now (default: every level the schedule visits), so
        that the next solve() is pure compute."""
        call("mgb_amg_prepare", self.handle, int(l))

    def level_size(self, l):
        N, nz = C.c_int(), C.c_int()
        call("mgb_amg_level_size", self.handle, l, C.byref(N), C.byref(nz))
        return N.value, nz.value

    def chol_info(self, l=None):
        """Device factorisation of level l (default finest): ranks it is split over, doubles exchanged and launches per
        Newton system, and whether the Hessian values stay on the rank that computed them (subtrees = row blocks)."""
        sw, ex, la = C.c_int(), C.c_double(), C.c_int()
        call("mgb_amg_chol_info", self.handle, self.L - 1 if l is None else int(l), C.byref(sw), C.byref(ex), C.byref(la))
        vl = C.c_int()
        call("mgb_amg_chol_values_local", self.handle, self.L - 1 if l is None else int(l), C.byref(vl))
        return dict(split_world=sw.value, exchange_doubles=ex.value, launches=la.value, values_local=bool(vl.value))

    def hessian_pattern(self, l):
        N, nz = self.level_size(l)
        rp = np.empty(N + 1, dtype=np.int32)
        ci = np.empty(nz, dtype=np.int32)
        call("mgb_amg_hessian_pattern", self.handle, l, iptr(rp), iptr(ci))
        return rp, ci

    def set_c(self, c):
        c = f64(c)
        assert c.shape == (self.n, self.K)
        call("mgb_amg_set_c", self.handle, dptr(c))

    def set_z(self, z):
        z = f64(np.asarray(z).reshape(-1))
        assert z.size == self.n * self.S
        call("mgb_amg_set_z", self.handle, dptr(z))

    def get_z(self):
        z = np.empty(self.n * self.S)
        call("mgb_amg_get_z", self.handle, dptr(z))
        return z

    def apply_D(self, l, s):
        s = f64(s)
        out = np.empty((self.n_local, self.K))
        call("mgb_amg_apply_D", self.handle, l, dptr(s), dptr(out))
        return out

    def apply_D_global(self, l, s):
        """apply_D on all n rows on every rank (a sharded AMG gathers the row blocks by summation)."""
        loc = self.apply_D(l, s)
        backend = self.geometry.x.backend
        if backend.world == 1:
            return loc
        full = np.zeros((self.n, self.K))
        full[self.row0:self.row0 + self.n_local] = loc
        v = HPCVector(full.reshape(-1), backend)
        call("mgb_vec_allreduce_sum", v.handle)
        return v.to_numpy().reshape(self.n, self.K)

    def f0(self, l, s, t, parts=False):
        s = f64(s)
        y = C.c_double()
        pr = np.empty(2)
        call("mgb_amg_f0", self.handle, l, dptr(s), float(t), C.byref(y), dptr(pr))
        return (y.value, pr) if parts else y.value

    def f0_trial(self, l, s_ref, s, t):
        """Line-search trial: f0(s), or +inf if a row lost more than 90 % of its cone distance w.r.t. s_ref."""
        s_ref, s = f64(s_ref), f64(s)
        y = C.c_double()
        call("mgb_amg_f0_trial", self.handle, l, dptr(s_ref), dptr(s), float(t), C.byref(y))
        return y.value

    def f1(self, l, s, t):
        s = f64(s)
        g = np.empty(self.level_size(l)[0])
        call("mgb_amg_f1", self.handle, l, dptr(s), float(t), dptr(g))
        return g

    def f2(self, l, s, t):
        """R'HR as a scipy CSR (full symmetric), assembled on the GPU."""
        s = f64(s)
        N, nz = self.level_size(l)
        vals = np.empty(nz)
        call("mgb_amg_f2", self.handle, l, dptr(s), float(t), dptr(vals))
        rp, ci = self.hessian_pattern(l)
        Lo = sp.csr_matrix((vals, ci, rp), shape=(N, N))
        return Lo + sp.tril(Lo, -1).T, vals

    # Float32 evaluation (csrc/kernels_f32.hip): the same SpMV / barrier kernels instantiated for float
    def f0_f32(self, l, s, t):
        s = np.ascontiguousarray(s, dtype=np.float32)
        y = C.c_double()
        call("mgb_amg_f0_f32", self.handle, l, s.ctypes.data_as(_lib.c_flt_p), float(t), C.byref(y))
        return y.value

    def f1_f32(self, l, s, t):
        s = np.ascontiguousarray(s, dtype=np.float32)
        g = np.empty(self.level_size(l)[0], dtype=np.float32)
        call("mgb_amg_f1_f32", self.handle, l, s.ctypes.data_as(_lib.c_flt_p), float(t), g.ctypes.data_as(_lib.c_flt_p))
        return g

    def f2_f32(self, l, s, t):
        """Lower-triangle values of R'HR in the order of hessian_pattern(l), float32."""
        s = np.ascontiguousarray(s, dtype=np.float32)
        vals = np.empty(self.level_size(l)[1], dtype=np.float32)
        call("mgb_amg_f2_f32", self.handle, l, s.ctypes.data_as(_lib.c_flt_p), float(t), vals.ctypes.data_as(_lib.c_flt_p))
        return vals

    def f1_template_f64(self, l, s, t):
        s = f64(s)
        g = np.empty(self.level_size(l)[0])
        call("mgb_amg_f1_template_f64", self.handle, l, dptr(s), float(t), dptr(g))
        return g

    def f2_template_f64(self, l, s, t):
        s = f64(s)
        vals = np.empty(self.level_size(l)[1])
        call("mgb_amg_f2_template_f64", self.handle, l, dptr(s), float(t), dptr(vals))
        return vals

    def f2_hpc(self, l, s, t) -> HPCSparseMatrix:
        """The Newton matrix R'HR of level l as an HPCSparseMatrix (what `f2` returns in the reference and what
        test/test_newton_matrix_compare.jl:33-51 captures); `.local_block(rank, world)` gives the per-rank fields."""
        H, _ = self.f2(l, s, t)
        return HPCSparseMatrix(H, self.geometry.x.backend)

    def solve_linear(self, l, lower_vals, g, solver="gpu"):
        """MultiGridBarrier.solve(A, b) = A \\ b on the level's fixed pattern: device (default) or host
        multifrontal Cholesky."""
        lower_vals, g = f64(lower_vals), f64(g)
        x = np.empty_like(g)
        call("mgb_amg_solve_linear_gpu" if solver == "gpu" else "mgb_amg_solve_linear", self.handle, l,
             dptr(lower_vals), dptr(g), dptr(x))
        return x

    SOLVERS = {"gpu": 0, "host": 1, "pcg": 2}

    def set_solver(self, solver="gpu"):
        """Newton linear solver: "gpu" = device multifrontal Cholesky (default), "host" = host Cholesky, "pcg" = conjugate
        gradients preconditioned by a V-cycle over the AMG levels with H applied matrix-free."""
        if solver not in self.SOLVERS:
            raise ValueError("solver must be 'gpu', 'host' or 'pcg'")
        call("mgb_amg_set_solver", self.handle, self.SOLVERS[solver])

    def set_pcg(self, rtol=0.0, maxit=0, degree=0, power_its=0, lo_frac=0.0, hi_frac=0.0, chunk=0, fallback=None,
                assembled_top=None, giveup=-1):
        """Parameters of solver="pcg" (unset ones keep their value): see mgb_amg_set_pcg."""
        flag = lambda v: -1 if v is None else int(bool(v))
        call("mgb_amg_set_pcg", self.handle, float(rtol), int(maxit), int(degree), int(power_its), float(lo_frac),
             float(hi_frac), int(chunk), flag(fallback), flag(assembled_top), int(giveup))

    # ---- multigrid pieces: mgb_hessian_apply / mgb_smooth / mgb_prolong / mgb_restrict
    def hessian_apply(self, l, s, v, matrix_free=True):
        """H(s) v at level l: matrix-free B' (Y o (B v)) on the device, or through the assembled matrix."""
        s, v = f64(s), f64(v)
        out = np.empty_like(v)
        call("mgb_hessian_apply", self.handle, l, dptr(s), dptr(v), dptr(out), int(bool(matrix_free)))
        return out

    def smooth(self, l, s, b, x0=None, degree=2, sweeps=1, lmax=0.0, matrix_free=True):
        """Chebyshev-Jacobi smoothing of H(s) x = b from x0 (zero by default); returns (x, lambda_max used)."""
        s, b = f64(s), f64(b)
        x = np.zeros_like(b) if x0 is None else f64(x0).copy()
        used = C.c_double()
        call("mgb_smooth", self.handle, l, dptr(s), dptr(b), dptr(x), int(degree), int(sweeps), float(lmax),
             int(bool(matrix_free)), C.byref(used))
        return x, used.value

    def prolong(self, l, xc):
        xc = f64(xc)
        xf = np.empty(self.level_size(l + 1)[0])
        call("mgb_prolong", self.handle, l, dptr(xc), dptr(xf))
        return xf

    def restrict(self, l, rf):
        rf = f64(rf)
        rc = np.empty(self.level_size(l)[0])
        call("mgb_restrict", self.handle, l, dptr(rf), dptr(rc))
        return rc

    def prolongation(self, l) -> sp.csr_matrix:
        """P_l (N_{l+1} x N_l) with R_l = R_{l+1} P_l."""
        r, c, nz = C.c_int(), C.c_int(), C.c_int()
        call("mgb_amg_prolongation", self.handle, l, C.byref(r), C.byref(c), C.byref(nz), None, None, None)
        rp, ci, va = np.empty(r.value + 1, dtype=np.int32), np.empty(nz.value, dtype=np.int32), np.empty(nz.value)
        call("mgb_amg_prolongation", self.handle, l, None, None, None, iptr(rp), iptr(ci), dptr(va))
        return sp.csr_matrix((va, ci, rp), shape=(r.value, c.value))

    def pcg_solve_linear(self, l, s, g):
        """x = H(s)^{-1} g by V-cycle-preconditioned CG; returns (x, iterations, relative residual in the M norm, converged)."""
        s, g = f64(s), f64(g)
        x = np.empty_like(g)
        it, ok, rr = C.c_int(), C.c_int(), C.c_double()
        call("mgb_amg_pcg_solve_linear", self.handle, l, dptr(s), dptr(g), dptr(x), C.byref(it), C.byref(rr), C.byref(ok))
        return x, it.value, rr.value, bool(ok.value)

    MG_KERNEL_NAMES = ("hessian_apply", "chebyshev_step", "hessian_apply_csr", "prolong", "restrict", "hessian_apply_unfused_csr")

    def time_mg_kernels(self, l, reps=50, nrot=1):
        """HIP-event timing of the multigrid kernels at level l (rotating operand copies as time_kernels)."""
        ms, by, alg = np.empty(6), np.empty(6), np.empty(6)
        call("mgb_amg_time_mg_kernels", self.handle, int(l), int(reps), int(nrot), dptr(ms), dptr(by), dptr(alg))
        return {k: dict(ms=float(a), bytes=float(b), algorithmic_bytes=float(c)) for k, a, b, c in zip(self.MG_KERNEL_NAMES, ms, by, alg)}

    def mg_coarsest(self, top=None):
        c0 = C.c_int()
        call("mgb_amg_mg_info", self.handle, self.L - 1 if top is None else int(top), C.byref(c0))
        return c0.value

    def solve(self, tol=None, t=0.1, kappa=10.0, maxit=10000, max_newton=0, verbose=0, schedule="fine",
              solver="gpu", stop_rule="fixed", centering="exact"):
        if schedule not in ("fine", "all"):
            raise ValueError("schedule must be 'fine' or 'all'")
        if stop_rule not in ("fixed", "upstream"):
            raise ValueError("stop_rule must be 'fixed' or 'upstream'")
        if centering not in ("decrement", "exact"):
            raise ValueError("centering must be 'decrement' or 'exact'")
        call("mgb_amg_set_stop_rule", self.handle, 1 if stop_rule == "upstream" else 0)
        call("mgb_amg_set_centering", self.handle, 1 if centering == "exact" else 0)
        call("mgb_amg_set_schedule", self.handle, 1 if schedule == "all" else 0)
        self.set_solver(solver)
        call("mgb_amg_solve", self.handle, float(tol or 0.0), float(t), float(kappa), int(maxit), int(max_newton),
             int(verbose))
        nt, te, tf = C.c_int(), C.c_double(), C.c_double()
        counts = (C.c_longlong * 4)()
        call("mgb_amg_sol_info", self.handle, C.byref(nt), C.byref(te), C.byref(tf), counts)
        its = np.empty(self.L * nt.value, dtype=np.int64)
        ts = np.empty(nt.value)
        cd = np.empty(nt.value)
        call("mgb_amg_sol_get", self.handle, its.ctypes.data_as(_lib.c_ll_p), dptr(ts), dptr(cd))
        nk = len(self.KERNEL_NAMES)
        kms, kby = np.empty(nk), np.empty(nk)
        kl = np.empty(nk, dtype=np.int64)
        call("mgb_amg_sol_kernels", self.handle, dptr(kms), dptr(kby), kl.ctypes.data_as(_lib.c_ll_p))
        kernels = {k: dict(ms=float(m), bytes=float(b), launches=int(c))
                   for k, m, b, c in zip(self.KERNEL_NAMES, kms, kby, kl)}
        pc = (C.c_longlong * 4)()
        tp = C.c_double()
        call("mgb_amg_sol_pcg", self.handle, pc, C.byref(tp))
        return dict(t_elapsed=te.value, ts=ts, its=its.reshape(nt.value, self.L).T.copy(), c_dot_Dz=cd,
                    time_factor=tf.value, n_f0=counts[0], n_f1=counts[1], n_f2=counts[2], n_factor=counts[3],
                    kernels=kernels, pcg=dict(solves=pc[0], iterations=pc[1], fallbacks=pc[2], gave_up_at=pc[3], seconds=tp.value))

    KERNEL_NAMES = ("apply_D", "barrier_f2", "hessian_assemble", "barrier_f1", "restrict", "barrier_f0",
                    "chol_front_start", "chol_front_step", "chol_backward_rect", "chol_backward", "chol_front_single")

    def time_kernels(self, l, reps=50, nrot=1):
        """Back-to-back launches of each kernel class; nrot > 1 rotates over that many distinct copies of every operand
        (working set nrot x bytes: beyond 256 MiB the rate is an HBM rate, not an Infinity-Cache rate)."""
        ms = np.empty(8)
        by = np.empty(8)
        call("mgb_amg_time_kernels", self.handle, l, reps, int(nrot), dptr(ms), dptr(by))
        names = ("apply_D", "barrier_f2", "hessian_assemble", "barrier_f1", "restrict", "barrier_f0", "trial_f0")
        out = {k: dict(ms=float(m), bytes=float(b)) for k, m, b in zip(names, ms, by)}
        out["apply_D"]["element_local"] = bool(by[7])
        out["apply_D_csr"] = dict(ms=float(ms[7]), bytes=float(by[0]))
        return out

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                _lib.load().mgb_amg_destroy(self.handle)
        except Exception:
            pass


def amg(geometry: Geometry, state_variables=DEFAULT_STATE, D=None, p=1.0, cones=None) -> AMG:
    return AMG(geometry, state_variables, D, p, cones=cones)


@dataclass
class AMGBSOL:
    """src:467-473 field order."""
    z: object
    SOL_feasibility: Optional[dict]
    SOL_main: dict
    log: list
    geometry: Geometry


PHASE1_SLACK_FLOOR = 1.0
PHASE1_PENALTY = 10.0


def _phase1_slack(geometry, M: "AMG", p, z0, c, tol, schedule, solver):
    """General feasibility phase (oracle amgb_phase1_slack; upstream amgb_phase1, SOL_feasibility src:428-455): the original
    problem relaxed by a slack field sigma (:full subspace, one more row `sigma id` of D) -- (q, s + sigma) in the power cone,
    coef . y + off + sigma > 0 for the half space, sigma > -1 -- with the penalty PHASE1_PENALTY max(1, |c|_max) on sigma,
    started strictly inside at sigma0 = 1 + the largest violation and path-followed on the device until sigma < 0 at every
    node (mgb_amg_set_early_stop).  Returns the strictly feasible (n, S) start of the main phase and the SOL fields."""
    n, S = z0.shape
    K = len(M.D)
    cone, extras = M.cones[0], list(M.cones[1:])
    if len(extras) > 1 or cone[0] == "linear":
        raise NotImplementedError("amgb: the feasibility phase covers the power cone intersected with one half space")
    state1 = tuple(M.state_variables) + (("sigma", "full"),)
    D1 = tuple(M.D) + (("sigma", "id"),)
    terms = [(list(cone[0]), cone[1], K)]
    terms += [("linear", list(e[1]) + [K], list(map(float, e[2])) + [1.0], float(e[3])) for e in extras]
    terms.append(("linear", [K], [1.0], PHASE1_SLACK_FLOOR))
    Dz = M.apply_D_global(M.L - 1, np.zeros(M.level_size(M.L - 1)[0]))
    idx = list(cone[0])
    viol = [np.sum(Dz[:, idx[:-1]] ** 2, axis=1) ** (cone[1] / 2.0) - Dz[:, idx[-1]]]
    viol += [-(Dz[:, list(e[1])] @ np.asarray(e[2], dtype=np.float64) + float(e[3])) for e in extras]
    sigma0 = 1.0 + max(0.0, float(np.max(np.concatenate(viol))))
    M1 = AMG(geometry, state1, D1, p, cones=terms)
    c1 = np.column_stack([c, np.full(n, PHASE1_PENALTY * max(1.0, float(np.max(np.abs(c)))))])
    M1.set_c(c1)
    M1.set_z(np.column_stack([z0, np.full(n, sigma0)]).reshape(-1, order="F"))
    call("mgb_amg_set_early_stop", M1.handle, K)
    M1.set_solver(solver)
    M1.prepare()
    SOL = M1.solve(tol=tol, schedule=schedule, solver=solver)
    z1 = M1.get_z().reshape((n, S + 1), order="F")
    if not np.max(z1[:, S]) < 0.0:
        raise MGBError(-3, "amgb: the problem is infeasible (the feasibility phase ended with a non-negative slack)")
    SOL["sigma0"] = sigma0
    return np.ascontiguousarray(z1[:, :S]), SOL


def _amgb_float32(geometry, M: "AMG", z0, tol, t, kappa, maxit, verbose):
    """Float32 main phase (the reference runs T = Float32 on its Metal backend, tolerance 1e-4: test/test_utils.jl:67-88,118-119;
    SURVEY.md section 8 f3).  The Newton loop of oracle amgb_core / newton / line search, driven from the host over the FLOAT
    instantiation of the device kernels -- objective, gradient and Hessian values through mgb_amg_f0_f32 / f1_f32 / f2_f32 -- with
    the Newton system solved by the double device Cholesky on the float-assembled values (there is no Float32 factorisation:
    fp64 runs at the fp32 vector rate on MI355X).  The iterate lives in double between centerings; s, Dz, the barrier terms, the
    gradient and the Hessian values are float.  Host-driven on purpose: the float path is for the small meshes the reference
    runs in Float32, not a hot path."""
    import time as _time
    l = M.L - 1
    R = sp.block_diag([geometry.subspaces[sv[1]][l].host for sv in M.state_variables], format="csr")
    N = R.shape[1]
    z = np.asarray(z0, dtype=np.float64).reshape(-1, order="F").copy()
    kappa0, t_begin = kappa, _time.time()
    t_stop = t
    while t_stop <= 1.0 / tol:
        t_stop *= kappa0
    max_newton, steps = 48, [0]

    def centre(zc, tc):
        M.set_z(zc)
        s = np.zeros(N, dtype=np.float32)
        y = M.f0_f32(l, s, tc)
        if not math.isfinite(y):
            return None
        g = M.f1_f32(l, s, tc).astype(np.float64)
        ymin, gmin = y, float(np.linalg.norm(g))
        for _ in range(max_newton):
            steps[0] += 1
            try:
                n = M.solve_linear(l, M.f2_f32(l, s, tc).astype(np.float64), g)
            except MGBError:
                return None
            inc = float(g @ n)
            if not math.isfinite(inc):
                return None
            if inc <= 0:
                return zc + R @ s.astype(np.float64)
            step, accepted = 1.0, False
            while step >= 1e-8:                               # backtracking: finite (amgb_all_isfinite, src:121) + Armijo
                st = (s.astype(np.float64) - step * n).astype(np.float32)
                yt = M.f0_f32(l, st, tc)
                if math.isfinite(yt) and yt <= y - 0.1 * step * inc:
                    gt = M.f1_f32(l, st, tc).astype(np.float64)
                    if np.all(np.isfinite(gt)):
                        accepted = True
                        break
                step *= 0.5
            if not accepted:
                yt, gt, st = y, g, s
            gn = float(np.linalg.norm(gt))
            done = yt >= ymin and gn >= 0.1 * gmin            # stagnation in float precision (oracle stopping_exact)
            s, y, g = st, yt, gt
            ymin, gmin = min(ymin, y), min(gmin, gn)
            if done:
                return zc + R @ s.astype(np.float64)
        return None

    def c_dot(zc):
        M.set_z(zc)
        return float(M.f0(l, np.zeros(N), 0.0, parts=True)[1][1])

    its, ts, cd = [], [], []
    zz = None
    for _ in range(8):                                        # INITIAL_CENTERING_ATTEMPTS
        zz = centre(z, t)
        if zz is not None:
            break
    if zz is None:
        raise MGBError(-3, "amgb (Float32): initial centering failed")
    z = zz
    its.append(steps[0]); ts.append(t); cd.append(c_dot(z))
    k = 1
    while t < t_stop and kappa > 1 and k < maxit:
        k += 1
        before = steps[0]
        while kappa > 1:
            t1 = min(kappa * t, t_stop)
            b1 = steps[0]
            zz = centre(z, t1)
            if zz is not None:
                if steps[0] - b1 <= max_newton * 0.25:
                    kappa = min(kappa0, kappa * kappa)
                z, t = zz, t1
                break
            kappa = math.sqrt(kappa)
            if kappa < 1 + 1e-3:
                kappa = 1.0
        its.append(steps[0] - before); ts.append(t); cd.append(c_dot(z))
        if verbose:
            print("[mgb f32] t=%.4g kappa=%.3g its=%d" % (t, kappa, its[-1]))
    if t < t_stop:
        raise MGBError(-3, "amgb (Float32): convergence failure (kappa collapsed)")
    itm = np.zeros((M.L, len(its)), dtype=np.int64)
    itm[l] = its
    M.set_z(z)
    return z.astype(np.float32).astype(np.float64), dict(t_elapsed=_time.time() - t_begin, ts=np.array(ts), its=itm,
                                                         c_dot_Dz=np.array(cd), T="float32")


def amgb(geometry: Geometry, p=1.0, state_variables=DEFAULT_STATE, D=None, f=None, g=None, tol=None, t=0.1,
         maxit=10000, kappa=10.0, verbose=False, logfile=None, schedule="fine", solver="gpu", cones=None, stop_rule="fixed",
         centering="exact", T=np.float64, **rest) -> AMGBSOL:
    """MultiGridBarrier.amgb on an MPI geometry (called at src:599,666).  kwargs as documented in
    docs/src/guide.md:148-152; unknown kwargs (e.g. `L`, forwarded by fem*d_mpi_solve, src:663-666)
    are ignored like Julia's `kwargs...` fan-out.  `cones` (upstream kwarg `Q`: the convex set) selects the barrier terms,
    see `AMG`; default = the p-Laplace power cone."""
    if geometry._geo is None:
        raise TypeError("amgb: geometry must come from native_to_mpi / fem*d_mpi")
    dim = geometry.discretization["dim"]
    f = DEFAULT_F[dim] if f is None else f
    g = DEFAULT_G[dim] if g is None else g
    M = AMG(geometry, state_variables, D, p, cones=cones, select=rest.get("select"))
    x = geometry.x.to_numpy()
    z0 = _rows(g, x)        # g_grid (n, S)
    c = _rows(f, x)         # f_grid (n, K)
    M.set_c(c)
    M.set_z(z0.reshape(-1, order="F"))
    Nf = M.level_size(M.L - 1)[0]
    y0 = M.f0(M.L - 1, np.zeros(Nf), 0.0)
    SOL_feasibility = None
    if not math.isfinite(y0) and rest.get("select") is not None:
        raise MGBError(-3, "amgb: a piecewise set (select=) needs a strictly feasible start")
    if not math.isfinite(y0):
        # Feasibility phase (SOL_feasibility, src:428-455).  For the power-cone family it has a closed form:
        # the slack row is `id` of a :full state variable (that space contains the constants), so a constant
        # shift sigma = 1 + max(|q|^p - s) of that variable is strictly feasible.  Dz comes from the device.
        idx = M.idx
        if idx and len(M.cones) > 1:
            # general feasibility phase: the set relaxed by a slack field, path-followed until the slack is negative
            z0, SOL_feasibility = _phase1_slack(geometry, M, p, z0, c, tol, schedule, solver)
            M.set_z(z0.reshape(-1, order="F"))
            if not math.isfinite(M.f0(M.L - 1, np.zeros(Nf), 0.0)):
                raise MGBError(-3, "amgb: feasibility phase failed")
        elif not idx:
            raise MGBError(-3, "amgb: infeasible start")
        else:
            var, op = M.D[idx[-1]]
            names = [sv[0] for sv in M.state_variables]
            if op != "id" or dict(M.state_variables)[var] != "full":
                raise NotImplementedError("amgb: feasibility phase needs the cone's slack to be `id` of a :full variable")
            Dz = M.apply_D_global(M.L - 1, np.zeros(Nf))
            q2 = np.sum(Dz[:, idx[:-1]] ** 2, axis=1)
            pv = M.p_nodes if M.p_nodes is not None else p
            sigma = 1.0 + float(np.max(q2 ** (pv / 2.0) - Dz[:, idx[-1]]))
            z0[:, names.index(var)] += sigma
            M.set_z(z0.reshape(-1, order="F"))
            if not math.isfinite(M.f0(M.L - 1, np.zeros(Nf), 0.0)):
                raise MGBError(-3, "amgb: feasibility phase failed")
            SOL_feasibility = dict(shift=sigma, its=np.zeros((M.L, 0), dtype=np.int64), ts=np.zeros(0),
                                   c_dot_Dz=np.zeros(0), t_elapsed=0.0)
    if np.dtype(T) == np.float32:                             # the reference's Float32 configurations (test/test_utils.jl:67-88)
        if SOL_feasibility is not None and "shift" not in SOL_feasibility:
            raise NotImplementedError("amgb: Float32 covers the closed-form feasibility shift only")
        M.prepare()
        tol32 = float(np.sqrt(np.finfo(np.float32).eps)) if tol is None else float(tol)
        z, SOL = _amgb_float32(geometry, M, z0, tol32, t, kappa, maxit, verbose)
        return AMGBSOL(HPCMatrix(z.reshape(z0.shape, order="F"), geometry.x.backend), SOL_feasibility, SOL, [], geometry)
    if np.dtype(T) != np.float64:
        raise ValueError("T must be float64 or float32")
    M.set_solver(solver)
    if rest.get("pcg"):
        M.set_pcg(**rest["pcg"])      # parameters of solver="pcg", see AMG.set_pcg
    M.prepare()       # factorisation structures are setup, not solve time (SOL_main.t_elapsed mirrors the reference's)
    SOL = M.solve(tol=tol, t=t, kappa=kappa, maxit=maxit, verbose=2 if verbose and verbose > 1 else int(bool(verbose)),
                  schedule=schedule, solver=solver, stop_rule=stop_rule, centering=centering)
    z = M.get_z().reshape(z0.shape, order="F")
    return AMGBSOL(HPCMatrix(z, geometry.x.backend), SOL_feasibility, SOL, [], geometry)


def fem1d_mpi_solve(L: int = 4, **kwargs) -> AMGBSOL:
    """src:594-600: kwargs go to both fem1d_mpi and amgb."""
    geo_kw = {k: kwargs[k] for k in ("Ti", "backend") if k in kwargs}
    return amgb(fem1d_mpi(L, **geo_kw), **{k: v for k, v in kwargs.items() if k not in geo_kw})


def fem3d_mpi_solve(L: int = 2, k: int = 3, D=None, f=None, g=None, **kwargs) -> AMGBSOL:
    """src:735-745: 3-D defaults D = [u id; u dx; u dy; u dz; s id], f = (.5,0,0,0,1), g = (|x|^2, 100)."""
    geo_kw = {kk: kwargs[kk] for kk in ("Ti", "backend") if kk in kwargs}
    rest = {kk: v for kk, v in kwargs.items() if kk not in geo_kw}
    return amgb(fem3d_mpi(L, k, **geo_kw), D=D or DEFAULT_D[3], f=f or DEFAULT_F[3], g=g or DEFAULT_G[3], **rest)


def fem2d_mpi_solve(L: int = 2, K=None, **kwargs) -> AMGBSOL:
    """src:661-667."""
    geo_kw = {k: kwargs[k] for k in ("Ti", "backend") if k in kwargs}
    return amgb(fem2d_mpi(L, K, **geo_kw), **{k: v for k, v in kwargs.items() if k not in geo_kw})


@dataclass
class ParabolicSOL:
    """src:512-516 field order: geometry, ts, u (one n x S snapshot per time step)."""
    geometry: Geometry
    ts: np.ndarray
    u: list


def parabolic_solve(geometry: Geometry, h=0.2, t0=0.0, t1=1.0, p=1.0, f1=None, g=None, tol=None, verbose=False,
                    schedule="fine", solver="gpu", **rest) -> ParabolicSOL:
    """MultiGridBarrier.parabolic_solve on an MPI geometry (imported at src:22,54; kwargs h, t1, p, verbose as in
    test/test_parabolic.jl:48 and docs/src/guide.md:367,377).  Implicit Euler for
        u_t - div(|grad u|^(p-2) grad u) = -f1 ;
    each step minimises int (1/2h)(s1 - 2 u u_k) + (1/p) s2 + f1 u subject to s1 >= u^2, s2 >= |grad u|^p with the
    barrier of the two-cone intersection (one GPU barrier solve per step; the time loop is host control flow).
    Dirichlet data = boundary trace of the initial condition g (time independent)."""
    if geometry._geo is None:
        raise TypeError("parabolic_solve: geometry must come from native_to_mpi / fem*d_mpi")
    dim = geometry.discretization["dim"]
    g = DEFAULT_G[dim] if g is None else g
    f1 = (lambda x: 0.5) if f1 is None else f1
    ops = ("dx", "dy", "dz")[:dim]
    state = (("u", "dirichlet"), ("s1", "full"), ("s2", "full"))
    D = (("u", "id"),) + tuple(("u", o) for o in ops) + (("s1", "id"), ("s2", "id"))
    K = dim + 3
    cones = [([0, K - 2], 2.0), (list(range(1, dim + 1)) + [K - 1], float(p))]
    M = AMG(geometry, state, D, p, cones=cones)
    x = geometry.x.to_numpy()
    n = x.shape[0]
    u0 = np.array([np.asarray(g(xi), dtype=np.float64).reshape(-1)[0] for xi in x])
    grad2 = sum((geometry.operators[o].host @ u0) ** 2 for o in ops)
    z = np.concatenate([u0, np.full(n, 1.0 + float(np.max(u0 * u0))),
                        np.full(n, 1.0 + float(np.max(grad2 ** (p / 2.0))))])
    fgrid = np.array([float(f1(xi)) for xi in x])
    nsteps = int(round((t1 - t0) / h))
    ts = t0 + h * np.arange(nsteps + 1)
    backend = geometry.x.backend
    u = [HPCMatrix(z.reshape(n, 3, order="F"), backend)]
    M.set_z(z)
    for _ in range(nsteps):
        c = np.zeros((n, K))
        c[:, 0] = fgrid - z[:n] / h
        c[:, K - 2] = 1.0 / (2.0 * h)
        c[:, K - 1] = 1.0 / p
        M.set_c(c)
        M.solve(tol=tol, verbose=int(bool(verbose)), schedule=schedule, solver=solver)
        z = M.get_z()
        u.append(HPCMatrix(z.reshape(n, 3, order="F"), backend))
    return ParabolicSOL(geometry, ts, u)


def mpi_to_native(obj):
    """src:355-517: gather device objects back to native numpy/scipy types."""
    if isinstance(obj, ParabolicSOL):                                   # src:495-517
        return ParabolicSOL(mpi_to_native(obj.geometry), obj.ts, [_to_cpu_array(uk) for uk in obj.u])
    if isinstance(obj, Geometry):
        conv = lambda m: m.to_scipy() if isinstance(m, HPCSparseMatrix) else m
        return Geometry(dict(obj.discretization), _to_cpu_array(obj.x), _to_cpu_array(obj.w),
                        {k: [conv(m) for m in v] for k, v in obj.subspaces.items()},
                        {k: conv(m) for k, m in obj.operators.items()},
                        [conv(m) for m in obj.refine], [conv(m) for m in obj.coarsen])
    if isinstance(obj, AMGBSOL):
        return AMGBSOL(_to_cpu_array(obj.z), obj.SOL_feasibility, obj.SOL_main, obj.log, mpi_to_native(obj.geometry))
    if isinstance(obj, (HPCVector, HPCMatrix)):
        return obj.to_numpy()
    if isinstance(obj, HPCSparseMatrix):
        return obj.to_scipy()
    return obj
