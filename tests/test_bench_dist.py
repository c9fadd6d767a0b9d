"""world_size-2 gloo test (CPU) of the only multi-rank logic this round has: bench.py runs one replica per
rank ("replicas only") and combines the ranks' timings as the contract demands (MAX over
ranks, whole-job aggregate value)."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    import bench
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    try:
        dist.barrier()
        elapsed = bench.max_over_ranks(1.0 + rank, dist, "cpu")        # rank 1 is the slow one
        value = bench.whole_job_value(224, 90, world, elapsed)
        q.put((rank, elapsed, value))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_two_rank_timing_reduction():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [o[1] for o in out] == [2.0, 2.0]                      # both ranks see the slowest rank's time
    assert out[0][2] == out[1][2] == pytest.approx(2 * 224 * 90 / 2.0)


def test_single_rank_is_identity():
    sys.path.insert(0, ROOT)
    import bench
    assert bench.max_over_ranks(3.5) == 3.5
    assert bench.whole_job_value(100, 10, 1, 2.0) == 500.0
