#!/usr/bin/env python3
"""bench.py -- fem2d p-Laplace multigrid-barrier solve on MI355X (BASELINE.json metric).

A "step" is one full `fem2d_mpi_solve`-equivalent main phase (amgb: t-continuation x level loop x
Newton) on a geometry + AMG hierarchy already resident in HBM.  value = n * (Newton steps) * K /
time  ("DoF/s per Newton step", BASELINE.md: DoF := n = rows of x, steps := sum(SOL_main.its)).
N>1: one process per GPU (torch.distributed over RCCL).  Default: each rank solves its own replica of the
workload (weak scaling, whole-job value); --shard: ONE solve row-block sharded over the ranks with RCCL
allreduce of the gradient / Hessian values and a replicated factorisation (strong scaling).

One JSON line on rank 0.  Extra objects: `roofline` (dominant HIP kernel, HIP-event timed inside the
solve on the library's own stream) and `cpu_baseline` (the numpy/scipy oracle timed on the host)."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: 8.0 TB/s spec


def cpu_baseline(L, p, budget_s):
    """Oracle ('port'): Newton steps of the SAME schedule the GPU path runs (finest subspace, t = 0.1, 1, 10, ...),
    on the same mesh, bounded to ~budget_s of host time."""
    import numpy as np
    import mgb_oracle as O
    # scipy.sparse products and SuperLU are single-threaded; pin the (few) dense BLAS calls to one thread too, so
    # that `cores` is the number of threads the sample really used
    import threadpoolctl
    limiter = threadpoolctl.threadpool_limits(limits=1)
    cores = 1
    g = O.fem2d(L)
    M = O.amg(g)
    x = M.x
    z = O.map_rows(lambda xi: O.DEFAULT_G[2](xi), x).reshape(-1, order="F")
    c = O.map_rows(lambda xi: O.DEFAULT_F[2](xi), x)
    B = O.Barrier(O.convex_Euclidian_power([1, 2, 3], p))
    R = M.R[-1]
    steps, t, t0 = 0, 0.1, time.time()
    while time.time() - t0 < budget_s:
        SOL = O.newton(lambda s, ref: B.f0_phi(s, x, M.w, t * c, R, M.D, z, ref),
                       lambda s: B.f1(s, x, M.w, t * c, R, M.D, z), lambda s: B.f2(s, x, M.w, t * c, R, M.D, z),
                       np.zeros(R.shape[1]), 4, O.stopping_exact(0.1))
        steps += SOL["k"]
        z = z + R @ SOL["x"]
        t *= 10.0
    dt = time.time() - t0
    limiter.restore_original_limits()
    n = x.shape[0]
    return dict(value=n * steps / dt, unit="DoF/s per Newton step", cores=int(cores), kind="port",
                sample="oracle/mgb_oracle.py (numpy/scipy, SuperLU solves): %d finest-level Newton steps (<=4 per "
                       "centering, t = 0.1, 1, 10, ...) on fem2d L=%d p=%g, %.1f s of host time" % (steps, L, p, dt))


def max_over_ranks(elapsed, dist=None, device="cpu"):
    """Slowest rank's wall time (the contract's MAX over ranks); identity when not distributed."""
    if dist is None:
        return float(elapsed)
    import torch
    tt = torch.tensor([float(elapsed)], dtype=torch.float64, device=device)
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    return float(tt.item())


def whole_job_value(n, newton_steps_per_rank, world, elapsed):
    """Replicas: every rank runs the same workload, so the job processed world * n * steps DoF-steps."""
    return world * n * newton_steps_per_rank / elapsed


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1)
    ap.add_argument("--warmup", type=int, default=0)
    ap.add_argument("--L", type=int, default=7)
    ap.add_argument("--p", type=float, default=1.0)
    ap.add_argument("--cpu-budget", type=float, default=15.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--verbose", type=int, default=0)
    ap.add_argument("--shard", action="store_true",
                    help="N>1: ONE solve row-block sharded over the ranks (RCCL allreduce of gradient / Hessian values, "
                         "replicated factorisation; strong scaling) instead of one replica per rank")
    ap.add_argument("--probe-L", type=int, default=9,
                    help="after the timed region, HIP-event time the barrier / SpMV kernels back to back on this larger "
                         "mesh (0 = skip): the size at which they leave the launch-latency regime")
    ap.add_argument("--rehearse-one-gpu", action="store_true",
                    help="N>1 on a single-GPU box: every rank uses cuda:0 and torch.distributed runs on gloo (RCCL refuses "
                         "two ranks on one device); exercises the multi-rank control flow, not a measurement")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    dist = None
    if world > 1:
        import torch.distributed as dist
        if args.rehearse_one_gpu:
            local_rank = 0
            torch.cuda.set_device(0)
            dist.init_process_group(backend="gloo")
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    import numpy as np
    import mgb_amd as M

    if M.device_count() <= 0:
        raise SystemExit("bench.py: no HIP device visible (the HIP path has no CPU fallback)")
    dev = local_rank if world > 1 else 0
    backend = M.backend_hip(dev)
    sharded = bool(args.shard and world > 1)
    if sharded:      # one solve over all ranks: row blocks + RCCL allreduce
        backend.set_comm(rank, world, M.torch_allreduce(dist, dev))

    # ---- setup (untimed): geometry upload + AMG hierarchy resident in HBM
    t_setup = time.time()
    geo = M.fem2d_mpi(args.L, backend=backend)
    A = M.AMG(geo, p=args.p)
    x = geo.x.to_numpy()
    z0 = np.vstack([M.DEFAULT_G[2](xi) for xi in x]).reshape(-1, order="F")
    c = np.vstack([M.DEFAULT_F[2](xi) for xi in x])
    A.set_c(c)
    A.prepare()                  # operators, Hessian plan, factorisation structures: setup, not solve
    t_setup = time.time() - t_setup
    n = A.n
    NL = A.level_size(A.L - 1)[0]

    def one_solve():
        A.set_z(z0)
        return A.solve(verbose=args.verbose)

    # preload code objects with a tiny solve (not a warmup step of the workload)
    small = M.AMG(M.fem2d_mpi(3 if sharded else 2, backend=backend), p=args.p)
    xs = small.geometry.x.to_numpy()
    small.set_c(np.vstack([M.DEFAULT_F[2](xi) for xi in xs]))
    small.set_z(np.vstack([M.DEFAULT_G[2](xi) for xi in xs]).reshape(-1, order="F"))
    small.solve()
    del small

    for _ in range(args.warmup):
        one_solve()

    def fence():
        backend.synchronize()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    fence()
    t0 = time.perf_counter()
    sols = [one_solve() for _ in range(args.steps)]
    backend.synchronize()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    elapsed = max_over_ranks(elapsed, dist, "cpu" if args.rehearse_one_gpu else "cuda")
    if dist is not None:
        dist.barrier()

    newton_steps = int(sum(int(s["its"].sum()) for s in sols))
    # replicas: every rank solved its own copy; sharded: all ranks worked on ONE solve
    value = whole_job_value(n, newton_steps, 1 if sharded else world, elapsed)
    if rank == 0:
        last = sols[-1]
        kern = {}
        for s in sols:
            for k, v in s["kernels"].items():
                d = kern.setdefault(k, dict(ms=0.0, bytes=0.0, launches=0))
                d["ms"] += v["ms"]; d["bytes"] += v["bytes"]; d["launches"] += v["launches"]
        # kernels are event-timed on every 8th Newton step only (bracketing every launch costs ~14 % of the solve)
        est = {k: v["ms"] * 8.0 for k, v in kern.items()}
        dom = max(est, key=lambda k: est[k])
        kd = kern[dom]
        achieved = kd["bytes"] / max(kd["ms"], 1e-12) / 1e6      # GB/s = bytes / ms / 1e6
        # HBM traffic per launch from the committed PMC passes (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE
        # runs of this same command, gfx950 FETCH_SIZE x2 correction applied); null for other workloads
        traffic = None
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "r1_pmc_traffic.json")))
            if pmc.get("workload") == "fem2d L=%d p=%g" % (args.L, args.p):
                traffic = pmc["kernels"][dom]["hbm_bytes_per_launch"]
        except Exception:
            traffic = None
        roofline = dict(bound="hbm", kernel=dom, achieved=achieved, peak=HBM_PEAK_GBS, unit="GB/s",
                        frac=achieved / HBM_PEAK_GBS, traffic=traffic,
                        avg_launch_us=1e3 * kd["ms"] / max(kd["launches"], 1), launches_timed=kd["launches"],
                        algorithmic_bytes_per_launch=kd["bytes"] / max(kd["launches"], 1),
                        note="latency-bound at this size: every working set is L2/Infinity-Cache resident and the "
                             "factorisation + solve is a dependent chain of 47 short launches (DESIGN.md sections 4b, 5)",
                        all_kernels={k: dict(gbs=v["bytes"] / max(v["ms"], 1e-12) / 1e6,
                                             avg_us=1e3 * v["ms"] / max(v["launches"], 1), launches_timed=v["launches"],
                                             est_total_s=est[k] / 1e3 / args.steps)
                                     for k, v in kern.items() if v["launches"]})
        out = {
            "metric": "fem2d p-Laplace DoF/s per Newton step", "value": value, "unit": "DoF/s per Newton step",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "strong" if sharded else "weak", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic" + (" (one-GPU rehearsal over gloo: not a measurement)" if args.rehearse_one_gpu else ""),
            "config": {"workload": "fem2d p-Laplace L=%d p=%g (n=%d rows, N_L=%d Newton unknowns), amgb main phase, "
                                   "tol=sqrt(eps)" % (args.L, args.p, n, NL),
                       "parallelism": "single GPU" if world == 1 else
                       ("row-block sharded x%d (RCCL allreduce of gradient + Hessian values, replicated factorisation)"
                        % world if sharded else "replicas x%d (not sharded)" % world)},
            "total_solve_s": elapsed / args.steps, "newton_steps_per_solve": newton_steps / args.steps,
            "linear_solver": "gpu multifrontal Cholesky (csrc/gpuchol.hip)",
            "linear_solve_s_per_solve": sum(s["time_factor"] for s in sols) / args.steps,
            "barrier_spmv_kernel_s_per_solve": sum(est[k] for k in kern if not k.startswith("chol_")) / 1e3 / args.steps,
            "setup_s": t_setup, "t_final": float(last["ts"][-1]), "c_dot_Dz_final": float(last["c_dot_Dz"][-1]),
            "roofline": roofline,
        }
        if sharded:
            out["allreduce"] = backend.comm_stats()
        if args.probe_L > 0 and world == 1:      # N=1 only, like the CPU baseline
            # secondary evidence for the bandwidth-shaped kernels: same kernels, back-to-back
            # launches, on the workload mesh (cache resident, launch bound) and on a mesh that exceeds L2
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import spmv_roofline
            out["kernel_bandwidth_probe"] = [spmv_roofline.probe(L, args.p) for L in sorted({args.L, args.probe_L})]
        if not args.no_cpu_baseline and world == 1:      # the contract: rank 0 at N=1 only
            out["cpu_baseline"] = cpu_baseline(args.L, args.p, args.cpu_budget)
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
