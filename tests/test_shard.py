"""Row-block sharding (reference: HPCSparseArrays row partition + MPI, src:216-221, 259-338).

CPU part (world_size-2 gloo): the host-only shard of a level plan through the C ABI -- each rank applies ITS
rows of B, ITS columns of BT and T, the ranks' contributions are summed with gloo all_reduce and must equal
the unsharded products (that is exactly what the GPU path does with the gradient and the Hessian values).
GPU part: two processes on the one GPU of the test box run the complete sharded solve, allreduce staged
through the host with gloo (RCCL refuses two ranks on one device), and must land on the unsharded solution."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

import mgb_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _full_plan(kind, L, level):
    """(geo handle, plan handle, n, S, K, nY, block, N, nnz) of the default problem, host-only."""
    from mgb_amd import _lib
    import mgb_amd as M
    import scipy.sparse as sp
    call, dptr, iptr, f64, i32 = _lib.call, _lib.dptr, _lib.iptr, _lib.f64, _lib.i32
    g = getattr(O, kind)(L)
    dim = g.discretization["dim"]
    x = f64(g.x.reshape(g.x.shape[0], -1))
    n = x.shape[0]
    block = {1: 2, 2: 7}[dim]
    h = C.c_void_p()
    call("mgb_geo_create", n, x.shape[1], len(g.refine), block, dptr(x), dptr(f64(g.w)), C.byref(h))

    def put(name, S):
        S = sp.csr_matrix(S)
        S.sort_indices()
        call("mgb_geo_set_matrix", h, name.encode(), S.shape[0], S.shape[1], iptr(i32(S.indptr)), iptr(i32(S.indices)),
             dptr(f64(S.data)))

    for k, S in g.operators.items():
        put("op:" + k, S)
    for k, v in g.subspaces.items():
        for l, S in enumerate(v):
            put("sub:%s:%d" % (k, l), S)
    state, D = M.DEFAULT_STATE, M.DEFAULT_D[dim]
    K = len(D)
    idx = list(range(K - dim - 1, K))
    iq = (C.c_int * (len(idx) - 1))(*idx[:-1])
    p = C.c_void_p()
    call("mgb_plan_create", h, len(state), _lib.str_array(state), K, _lib.str_array(D), len(idx) - 1, iq, idx[-1], level,
         C.byref(p))
    N, nz, nT, nB = (C.c_int() for _ in range(4))
    call("mgb_plan_sizes", p, C.byref(N), C.byref(nz), C.byref(nT), C.byref(nB))
    return h, p, n, len(state), K, block, N.value, nz.value


def _shard_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    for pth in (ROOT, os.path.join(ROOT, "oracle")):
        sys.path.insert(0, pth)
    import torch
    import torch.distributed as dist
    from mgb_amd import _lib
    call, dptr, f64 = _lib.call, _lib.dptr, _lib.f64
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    try:
        out = {}
        for kind, L in (("fem1d", 3), ("fem2d", 2)):
            level = L - 1
            h, p, n, S, K, block, N, nz = _full_plan(kind, L, level)
            ps = C.c_void_p()
            r0, r1 = C.c_int(), C.c_int()
            call("mgb_plan_shard", p, S, K, rank, world, block, C.byref(ps), C.byref(r0), C.byref(r1))
            nl = r1.value - r0.value
            assert nl % block == 0 and nl > 0
            rng = np.random.default_rng(7)                    # same stream on every rank
            s = rng.standard_normal(N)
            v = rng.standard_normal(n * K)
            # B: local rows of the full product
            Bs_full = np.empty(n * K)
            call("mgb_plan_apply_B_host", p, dptr(f64(s)), dptr(Bs_full))
            Bs = np.empty(nl * K)
            call("mgb_plan_apply_B_host", ps, dptr(f64(s)), dptr(Bs))
            assert np.array_equal(Bs, Bs_full[r0.value * K:r1.value * K])
            # BT: contributions of the row blocks sum to the full restriction
            g_full = np.empty(N)
            call("mgb_plan_apply_BT_host", p, dptr(f64(v)), dptr(g_full))
            g = np.empty(N)
            call("mgb_plan_apply_BT_host", ps, dptr(f64(v[r0.value * K:r1.value * K].copy())), dptr(g))
            tg = torch.from_numpy(g)
            dist.all_reduce(tg)
            assert np.abs(g - g_full).max() <= 1e-13 * np.abs(g_full).max()
            # T: Hessian values
            nT = C.c_int()
            call("mgb_plan_sizes", ps, None, None, C.byref(nT), None)
            dim = 1 if kind == "fem1d" else 2
            nY = (dim + 1) * (dim + 2) // 2
            Y = rng.standard_normal((n, nY))
            a_full = np.empty(nz)
            call("mgb_plan_eval_host", p, dptr(f64(Y)), dptr(a_full))
            a = np.empty(nz)
            call("mgb_plan_eval_host", ps, dptr(f64(Y[r0.value:r1.value].copy())), dptr(a))
            ta = torch.from_numpy(a)
            dist.all_reduce(ta)
            assert np.abs(a - a_full).max() <= 1e-13 * np.abs(a_full).max()
            out[kind] = (r0.value, r1.value, n)
            call("mgb_plan_destroy", ps)
            call("mgb_plan_destroy", p)
            call("mgb_geo_destroy", h)
        q.put((rank, out))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def _run(target, world, extra=()):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() * 7 + 13) % 2000
    procs = [ctx.Process(target=target, args=(r, world, port, q) + tuple(extra)) for r in range(world)]
    for p in procs:
        p.start()
    out = sorted((q.get(timeout=600) for _ in range(world)), key=lambda o: o[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    return out


def test_shard_rows_are_element_aligned_and_cover(lib):
    for n, block, world in ((16, 2, 2), (224, 7, 2), (57344, 7, 8), (224, 7, 3)):
        prev = 0
        for r in range(world):
            r0, r1 = C.c_int(), C.c_int()
            assert lib.mgb_shard_rows(r, world, n, block, C.byref(r0), C.byref(r1)) == 0
            assert r0.value == prev and r1.value > r0.value and r0.value % block == 0 and r1.value % block == 0
            prev = r1.value
        assert prev == n
    r0, r1 = C.c_int(), C.c_int()
    assert lib.mgb_shard_rows(0, 4, 14, 7, C.byref(r0), C.byref(r1)) != 0      # fewer elements than ranks


def test_plan_shards_sum_to_the_full_plan_gloo_world2():
    out = _run(_shard_worker, 2)
    (ra, a), (rb, b) = out
    for kind in a:
        assert a[kind][0] == 0 and a[kind][1] == b[kind][0] and b[kind][1] == a[kind][2]     # a partition


def _gpu_worker(rank, world, port, q, kind, L, p):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    for pth in (ROOT, os.path.join(ROOT, "oracle")):
        sys.path.insert(0, pth)
    import torch.distributed as dist
    import mgb_amd as M
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    try:
        be = M.backend_hip(0)
        be.set_comm(rank, world, M.torch_allreduce(dist, 0))
        sol = getattr(M, kind + "_mpi_solve")(L=L, p=p)
        z = M.mpi_to_native(sol).z
        info = M.AMG(getattr(M, kind + "_mpi")(L, backend=be), p=p).chol_info()
        q.put((rank, z, np.asarray(sol.SOL_main["its"]), dict(be.comm_stats(), **info)))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("kind,L,p", [("fem1d", 4, 1.0), ("fem2d", 3, 1.5), ("fem2d", 5, 1.5)])   # L=5: hipGraph replay path
def test_sharded_solve_matches_unsharded_two_processes_one_gpu(gpu_required, kind, L, p):
    import mgb_amd as M
    ref = M.mpi_to_native(getattr(M, kind + "_mpi_solve")(L=L, p=p)).z
    out = _run(_gpu_worker, 2, (kind, L, p))
    (r0, z0, its0, st0), (r1, z1, its1, st1) = out
    assert np.array_equal(z0, z1) and np.array_equal(its0, its1)          # ranks stay in lock step, bit for bit
    assert st0["calls"] == st1["calls"] > 0
    # the factorisation itself is split over the two ranks (one nested-dissection subtree each, replicated top)
    # (the top of the tree follows the row partition, so even fem1d L=4 with its 32 unknowns splits)
    assert st0["split_world"] == st1["split_world"] == 2 and st0["exchange_doubles"] > 0
    assert st0["values_local"] and st1["values_local"]
    assert np.linalg.norm(z0 - ref) <= 1e-10 * np.linalg.norm(ref)


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 4])
def test_sharded_hessian_values_stay_on_their_rank(gpu_required, world):
    """Reduce-to-owner (row (e)): the factorisation's subtrees follow the row partition, so the all-nnz allreduce of the
    Hessian values per Newton step is gone -- only entries among separator unknowns travel, inside the Schur-complement
    collective.  Same z as the unsharded solve; against MGB_RANK_ALIGNED=0 (geometric tree + full allreduce, the previous
    scheme) one collective and 40-60 % of the bytes per Newton step disappear."""
    import mgb_amd as M
    kind, L, p = "fem2d", 5, 1.5
    ref = M.mpi_to_native(M.fem2d_mpi_solve(L=L, p=p)).z
    res = {}
    for mode in ("1", "0"):
        os.environ["MGB_RANK_ALIGNED"] = mode
        try:
            out = _run(_gpu_worker, world, (kind, L, p))
        finally:
            del os.environ["MGB_RANK_ALIGNED"]
        zs, its, sts = [o[1] for o in out], [o[2] for o in out], [o[3] for o in out]
        assert all(np.array_equal(z, zs[0]) for z in zs) and all(np.array_equal(i, its[0]) for i in its)
        assert np.linalg.norm(zs[0] - ref) <= 1e-10 * np.linalg.norm(ref)
        assert all(st["split_world"] == world and st["values_local"] == (mode == "1") for st in sts)
        steps = int(its[0].sum())
        res[mode] = (sts[0]["calls"] / steps, sts[0]["bytes"] / steps)
    print("world %d: collectives / bytes per Newton step: aligned %.1f / %.0f, allreduce of all values %.1f / %.0f" % (
        world, *res["1"], *res["0"]))
    assert res["1"][0] < res["0"][0] - 0.5        # one collective per factorisation less (trial counts differ a little)
    assert res["1"][1] < 0.7 * res["0"][1]        # L=5: 41-60 % fewer bytes; at L=7 x 8 ranks 6.5 -> 2.9 MB per step


def _nccl_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    import mgb_amd as M
    torch.cuda.set_device(0)
    dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
    try:
        fn = M.torch_allreduce(dist, 0)
        t = torch.arange(1000, dtype=torch.float64, device="cuda:0")
        fn(t.data_ptr(), t.numel())                 # RCCL all_reduce in place on a raw device pointer
        q.put((rank, float(t.sum().item())))
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_rccl_allreduce_callback_on_a_raw_device_pointer(gpu_required):
    """The production callback (torch.distributed backend "nccl" = RCCL) on the one GPU of the test box: a
    world of one rank, so the sum is the identity -- what is exercised is the raw-pointer wrapping, the RCCL
    call and the stream hand-over that bench.py --shard relies on."""
    out = _run(_nccl_worker, 1)
    assert out[0][1] == 999 * 1000 / 2


def _lib_rccl_worker(rank, world, port, q):
    os.environ.update(HSA_ENABLE_IPC_MODE_LEGACY="0")
    sys.path.insert(0, ROOT)
    import numpy as np
    import mgb_amd as M
    from mgb_amd import _lib
    be = M.HPCBackend(0)
    be.set_comm_rccl(0, 1, M.rccl_unique_id())      # ncclCommInitRank inside the library, no torch in this process
    v = M.HPCVector(np.arange(1000.0), be)
    _lib.call("mgb_vec_allreduce_sum", v.handle)    # ncclAllReduce on the context stream
    st = be.comm_stats()
    # a full solve on a context that carries the communicator (world 1: the Newton loop itself never reduces)
    sol = M.fem2d_mpi_solve(L=2, p=1.5, backend=be)
    q.put((rank, float(v.to_numpy().sum()), st["calls"], st["bytes"], float(np.linalg.norm(M.mpi_to_native(sol).z))))


@pytest.mark.gpu
def test_library_owned_rccl_communicator_one_rank(gpu_required):
    """mgb_rccl_unique_id / mgb_ctx_set_comm_rccl: the communicator lives in the library (the reference's collectives are
    in-library too, src:125,132) and the collective is an ncclAllReduce enqueued on the context stream.  The test box has
    one GPU and RCCL refuses two ranks on one device, so the world has one rank: what is exercised is librccl's run-time
    binding, ncclCommInitRank, the stream-ordered collective and the teardown."""
    out = _run(_lib_rccl_worker, 1)
    assert out[0][1] == 999 * 1000 / 2 and out[0][2] == 1 and out[0][3] == 8000.0 and out[0][4] > 0


def _rccl_from_torch_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    sys.path.insert(0, ROOT)
    import numpy as np
    import torch
    import torch.distributed as dist
    import mgb_amd as M
    from mgb_amd import _lib
    torch.cuda.set_device(0)
    dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
    try:
        be = M.HPCBackend(0)
        M.rccl_comm_from_torch(be, dist)            # torch (with its own RCCL) only carries the id; the library opens librccl itself
        v = M.HPCVector(np.arange(100.0), be)
        _lib.call("mgb_vec_allreduce_sum", v.handle)
        t = torch.ones(4, device="cuda:0")
        dist.all_reduce(t)                          # torch's communicator still works beside the library's
        q.put((rank, float(v.to_numpy().sum()), float(t.sum().item()), be.comm_stats()["calls"]))
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_library_rccl_beside_torch_distributed(gpu_required):
    """bench.py's multi-GPU setup in one process: torch.distributed (backend nccl, its bundled RCCL) broadcasts the unique id, the
    library binds librccl at run time and owns its communicator; both communicators work side by side (world of one rank: the
    box has one GPU)."""
    out = _run(_rccl_from_torch_worker, 1)
    assert out[0][1] == 99 * 100 / 2 and out[0][2] == 4.0 and out[0][3] == 1


def test_plan_shards_sum_to_the_full_plan_gloo_world3_uneven():
    """Three ranks: 8 / 32 elements do not divide evenly, the row blocks differ in size and still partition."""
    out = _run(_shard_worker, 3)
    for kind in out[0][1]:
        blocks = [o[1][kind] for o in out]
        assert blocks[0][0] == 0 and blocks[-1][1] == blocks[0][2]
        assert all(a[1] == b[0] for a, b in zip(blocks, blocks[1:]))
        assert len({b[1] - b[0] for b in blocks}) > 1          # uneven


# ---------------------------------------------------------------- distributed factorisation (subtrees -> ranks)
def _chol_dist_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    for pth in (ROOT, os.path.join(ROOT, "oracle")):
        sys.path.insert(0, pth)
    import torch
    import torch.distributed as dist
    from mgb_amd import _lib
    call, dptr, f64 = _lib.call, _lib.dptr, _lib.f64
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    try:
        calls = []

        def thunk(_user, ptr, count):                       # sum-allreduce of HOST doubles in place (gloo)
            a = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_double)), shape=(count,))
            t = torch.from_numpy(a)
            dist.all_reduce(t)
            calls.append(int(count))
            return 0

        cb = _lib.ALLREDUCE_FN(thunk)
        out = {}
        for kind, L in ((("fem2d", 4), ("fem1d", 7)) if world < 8 else (("fem2d", 5), ("fem1d", 10))):      # 8 subtrees need a deeper tree
            h, p, n, S, K, block, N, nz = _full_plan(kind, L, L - 1)
            dim = 1 if kind == "fem1d" else 2
            nY = (dim + 1) * (dim + 2) // 2
            rng = np.random.default_rng(11)                  # same stream on every rank
            # an SPD Newton matrix from the plan: Y = w-like positive diagonal slots, small off-diagonal ones
            Y = np.zeros((n, nY))
            slot = 0
            for a in range(dim + 1):
                for b in range(a, dim + 1):
                    Y[:, slot] = (1.0 + rng.random(n)) if a == b else 0.1 * rng.standard_normal(n)
                    slot += 1
            vals = np.empty(nz)
            call("mgb_plan_eval_host", p, dptr(f64(Y)), dptr(vals))
            g = rng.standard_normal(N)
            ch = C.c_void_p()
            call("mgb_plan_hostchol_create", p, dim, C.byref(ch))
            x_ref = np.empty(N)
            call("mgb_hostchol_factor_solve", ch, dptr(vals), dptr(f64(g)), dptr(x_ref))
            sw, nn = C.c_int(), C.c_int()
            call("mgb_hostchol_partition", ch, world, C.byref(sw), 0, C.byref(nn), None)
            owner = np.empty(nn.value, dtype=np.int32)
            call("mgb_hostchol_partition", ch, world, C.byref(sw), nn.value, C.byref(nn), owner.ctypes.data_as(_lib.c_int_p))
            x = np.empty(N)
            n0 = len(calls)
            call("mgb_hostchol_factor_solve_dist", ch, rank, world, cb, None, dptr(vals), dptr(f64(g)), dptr(x))
            out[kind] = dict(split=sw.value, owner=owner, err=float(np.abs(x - x_ref).max() / np.abs(x_ref).max()),
                             x=x, exchanges=calls[n0:])
            # a non-SPD matrix on the subtree of ONE rank must fail on every rank (flag travels with the exchange)
            bad = vals.copy()
            rp = np.empty(N + 1, dtype=np.int32)
            ci = np.empty(nz, dtype=np.int32)
            call("mgb_plan_pattern", p, _lib.iptr(rp), _lib.iptr(ci))
            diag = [k for r in range(N) for k in range(rp[r], rp[r + 1]) if ci[k] == r]
            bad[diag[0]] = -1.0
            rc = _lib.load().mgb_hostchol_factor_solve_dist(ch, rank, world, cb, None, dptr(bad), dptr(f64(g)), dptr(x.copy()))
            out[kind]["bad_rc"] = rc
            # reduce-to-owner: the tree's top follows the row partition, every rank passes only ITS row block's contributions
            # to the matrix entries; the entries among separator unknowns ride in the first collective
            chr_ = C.c_void_p()
            call("mgb_plan_hostchol_create_ranked", p, dim, K, world, block, C.byref(chr_))
            al, ntop = C.c_int(), C.c_int()
            call("mgb_hostchol_rank_aligned", chr_, world, C.byref(al), C.byref(ntop))
            ps = C.c_void_p()
            r0, r1 = C.c_int(), C.c_int()
            call("mgb_plan_shard", p, S, K, rank, world, block, C.byref(ps), C.byref(r0), C.byref(r1))
            a_loc = np.empty(nz)
            call("mgb_plan_eval_host", ps, dptr(f64(Y[r0.value:r1.value].copy())), dptr(a_loc))
            xl = np.empty(N)
            n1 = len(calls)
            call("mgb_hostchol_factor_solve_dist_local", chr_, rank, world, cb, None, dptr(a_loc), dptr(f64(g)), dptr(xl))
            sw2 = C.c_int()
            call("mgb_hostchol_partition", chr_, world, C.byref(sw2), 0, C.byref(nn), None)
            out[kind].update(aligned=al.value, ntop=ntop.value, nz=nz, split_ranked=sw2.value, x_local=xl,
                             err_local=float(np.abs(xl - x_ref).max() / np.abs(x_ref).max()), exchanges_local=calls[n1:],
                             local_nonzero=int(np.count_nonzero(a_loc)))
            call("mgb_plan_destroy", ps)
            call("mgb_hostchol_destroy", chr_)
            call("mgb_hostchol_destroy", ch)
            call("mgb_plan_destroy", p)
            call("mgb_geo_destroy", h)
        q.put((rank, out))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4, 8])      # 8 = BASELINE configs[2] (8x row-partitioned): three levels of guided splits
def test_distributed_factor_solve_host_gloo(world):
    """The factorisation split by nested-dissection subtrees (one subtree per rank, replicated top, Schur complements and
    right-hand-side updates of the subtree roots summed over the ranks -- the scheme csrc/gpuchol.hip runs on the GPUs)
    through the C ABI on `world` gloo ranks: x equals the undistributed solve, the ranks agree bit for bit, exactly two
    collectives per solve, and a non-SPD matrix is reported by every rank."""
    out = _run(_chol_dist_worker, world)
    for kind in out[0][1]:
        rs = [o[1][kind] for o in out]
        assert all(r["split"] == world for r in rs)
        own = rs[0]["owner"]
        assert set(own.tolist()) == set(range(-1, world))              # every rank owns a subtree, the top is replicated
        assert own[-1] == -1                                           # the root (last in postorder) is top
        for r in rs:
            assert np.array_equal(r["owner"], own)
            assert r["err"] < 1e-12
            assert np.array_equal(r["x"], rs[0]["x"])
            assert len(r["exchanges"]) == 2 and r["exchanges"] == rs[0]["exchanges"]
            assert r["bad_rc"] == -3                                   # MGB_E_NUMERIC everywhere
            # reduce-to-owner (tree top = row partition): same solution from rank-local matrix entries, still two collectives,
            # and the only matrix entries that travel are the ones among separator unknowns -- a small fraction of nnz
            assert r["aligned"] == 1 and r["split_ranked"] == world
            assert r["err_local"] < 1e-11
            assert np.array_equal(r["x_local"], rs[0]["x_local"])
            assert len(r["exchanges_local"]) == 2 and r["exchanges_local"] == rs[0]["exchanges_local"]
            assert 0 < r["ntop"] < (0.2 if kind == "fem2d" else 0.05) * r["nz"]
            assert r["local_nonzero"] < r["nz"]                          # a rank's shard really does not see every entry


def test_partition_falls_back_to_replication(lib):
    """world that is not a power of two, or larger than the tree can be split into: everything replicated (split = 1)."""
    from mgb_amd import _lib
    h, p, n, S, K, block, N, nz = _full_plan("fem2d", 2, 1)
    ch = C.c_void_p()
    _lib.call("mgb_plan_hostchol_create", p, 2, C.byref(ch))
    for world in (1, 3, 6, 1024):
        sw, nn = C.c_int(), C.c_int()
        _lib.call("mgb_hostchol_partition", ch, world, C.byref(sw), 0, C.byref(nn), None)
        assert sw.value == 1
    _lib.call("mgb_hostchol_destroy", ch)
    _lib.call("mgb_plan_destroy", p)
    _lib.call("mgb_geo_destroy", h)
