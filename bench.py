#!/usr/bin/env python3
"""bench.py -- fem2d p-Laplace multigrid-barrier solve on MI355X (BASELINE.json metric).

A "step" is one full `fem2d_mpi_solve`-equivalent main phase (amgb: t-continuation x level loop x Newton) on a
geometry + AMG hierarchy already resident in HBM.  value = n * (Newton steps of the timed solves) / time
("DoF/s per Newton step", BASELINE.md: DoF := n = rows of x, steps := sum(SOL_main.its)).  The reference's own
benchmark times the whole `fem2d_mpi_solve` call including the hierarchy build (tools/benchmark_fem2d.jl:70-79), so
the line also carries `setup_s` and `total_solve_s_incl_setup` = setup + one solve.

N>1 (`--gpus N`): one process per GPU (torch.distributed over RCCL), launched by the driver through torchrun or -- when
WORLD_SIZE is not set -- spawned from here before anything touches the GPU.  Default: ONE solve, row-block sharded over
the ranks (BASELINE.json configs[2]: rows of x / Dz / the barrier kernels / the operators split by element blocks, the
factorisation split by nested-dissection subtrees with a replicated top, RCCL allreduce for the exchanges; strong
scaling).  `--replicas`: every rank solves its own copy (throughput mode, weak scaling).

One JSON line on rank 0.  Extra objects: `roofline` (dominant HIP kernel, HIP-event timed inside the solve on the
library's own stream), `cpu_baseline` (the same Newton path on the host cores), `parity` (z against the committed oracle
vector of this workload, when there is one)."""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: 8.0 TB/s spec


def cpu_baseline(L, p, budget_s):
    """The Newton path of this workload on the HOST cores, natively (`kind = "port"`): oracle/cpu/mgb_cpu_newton.cpp, a C++ /
    OpenMP restatement of the oracle's amgb_core / newton / line search / barrier f0-f1-f2 (pinned against the numpy oracle by
    tests/test_oracle_kats.py) on all cores the process may use, with the Hessian plan and the multifrontal Cholesky of the
    product's host code (csrc/amg.cpp build_level_plan, csrc/mfchol.cpp; threaded).  Same mesh, same problem, same stopping
    rules; stopped after the first centering that ends beyond the budget (~budget_s of host time)."""
    import ctypes as C
    import __graft_entry__ as G
    lib = C.CDLL(G.build_cpu_port())      # prebuilt by build(); compiled here only if missing or stale
    dp, ip, lp = C.POINTER(C.c_double), C.POINTER(C.c_int), C.POINTER(C.c_longlong)
    lib.mgb_cpu_solve.argtypes = [C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int, dp, lp, dp, dp, ip]
    steps, sec, t_reached, threads = C.c_longlong(), C.c_double(), C.c_double(), C.c_int()
    rc = lib.mgb_cpu_solve(2, int(L), 3, float(p), float(budget_s), 0, None, C.byref(steps), C.byref(sec), C.byref(t_reached),
                           C.byref(threads))
    if rc != 0:
        raise RuntimeError("cpu_baseline: the C++ port failed (see stderr)")
    n = 14 * 4 ** (int(L) - 1)
    return dict(value=n * steps.value / sec.value, unit="DoF/s per Newton step", cores=int(threads.value), kind="port",
                sample="C++ / OpenMP restatement of the same Newton path (oracle/cpu/mgb_cpu_newton.cpp: t-continuation, finest-level "
                       "Newton, line search, barrier f0 / f1 / f2, threaded Hessian plan + host multifrontal Cholesky of the product) "
                       "on %d OpenMP threads (the multifrontal Cholesky inside uses the product's own pool of at most 16): the first %d Newton steps of the same solve (t = 0.1 ... %.3g, stopped after the "
                       "centering that passed the budget) on fem2d L=%d p=%g, %.1f s of host time"
                       % (threads.value, steps.value, t_reached.value, L, p, sec.value))


def max_over_ranks(elapsed, dist=None, device="cpu"):
    """Slowest rank's wall time (the contract's MAX over ranks); identity when not distributed."""
    if dist is None:
        return float(elapsed)
    import torch
    tt = torch.tensor([float(elapsed)], dtype=torch.float64, device=device)
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    return float(tt.item())


def whole_job_value(n, newton_steps_per_rank, world, elapsed):
    """Replicas: every rank runs the same workload, so the job processed world * n * steps DoF-steps.  A sharded job
    (all ranks on ONE solve) passes world = 1."""
    return world * n * newton_steps_per_rank / elapsed


def spawn_ranks(ngpus, argv):
    """`python bench.py --gpus N` without a launcher: start the N ranks as fresh child processes (torchrun) BEFORE this
    process touches the GPU, relay their output and exit with their code."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ngpus), "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    return subprocess.call(cmd)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1)
    ap.add_argument("--warmup", type=int, default=0)
    ap.add_argument("--L", type=int, default=7)
    ap.add_argument("--p", type=float, default=1.0)
    ap.add_argument("--cpu-budget", type=float, default=15.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--verbose", type=int, default=0)
    ap.add_argument("--replicas", action="store_true",
                    help="N>1: one independent replica of the workload per rank (throughput mode, weak scaling) instead of "
                         "ONE row-block sharded solve over all ranks (the default, BASELINE.json configs[2])")
    ap.add_argument("--shard", action="store_true", help="(default for N>1; kept for compatibility)")
    ap.add_argument("--probe-L", type=int, default=9,
                    help="after the timed region, HIP-event time the barrier / SpMV kernels on this larger mesh with rotating "
                         "buffers (0 = skip): the size at which they leave the launch-latency regime")
    ap.add_argument("--solver", choices=("gpu", "pcg"), default="gpu",
                    help="Newton linear solver: device multifrontal Cholesky (default) or V-cycle-preconditioned CG with the "
                         "matrix-free Hessian (single GPU; DESIGN.md section 4c says where each wins)")
    ap.add_argument("--comm", choices=("rccl", "callback"), default="rccl",
                    help="sharded runs: collectives through the library-owned RCCL communicator on the context stream (default) or "
                         "through the torch.distributed callback (what --rehearse-one-gpu needs: RCCL refuses two ranks on one GPU)")
    ap.add_argument("--rehearse-one-gpu", action="store_true",
                    help="N>1 on a single-GPU box: every rank uses cuda:0 and torch.distributed runs on gloo (RCCL refuses "
                         "two ranks on one device); exercises the multi-rank control flow, not a measurement")
    return ap.parse_args(argv)


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))      # nothing has touched the GPU yet
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        sys.exit("bench.py: --gpus %d but the launcher started %d rank(s)" % (args.gpus, world))
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
    sharded = bool(world > 1 and not args.replicas)
    if sharded:      # one solve over all ranks: row blocks + RCCL allreduce
        use_rccl = args.comm == "rccl" and not args.rehearse_one_gpu
        if use_rccl:
            # the library owns the communicator; torch only carries the 128-byte id.  All ranks agree on whether it came up: if
            # any of them could not open librccl / initialise its rank, every rank drops back to the callback path together
            ok = 1
            try:
                M.rccl_comm_from_torch(backend, dist)
            except Exception as exc:
                print("bench.py: library-owned RCCL communicator unavailable on rank %d (%r)" % (rank, exc), file=sys.stderr)
                ok = 0
            flag = torch.tensor([ok], dtype=torch.int32, device=torch.device("cuda", dev))
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            use_rccl = bool(flag.item())
        if not use_rccl:
            backend.set_comm(rank, world, M.torch_allreduce(dist, dev))
    if sharded and args.solver != "gpu":
        raise SystemExit("bench.py: --solver pcg runs on one GPU")

    # preload code objects and the HIP context with a tiny solve (not a warmup step of the workload): `setup_s` below is the
    # hierarchy build of a warm process, as the reference's benchmark times its larger meshes after the smaller ones have
    # compiled everything (tools/benchmark_fem2d.jl:46-79)
    small = M.AMG(M.fem2d_mpi(3 if sharded else 2, backend=backend), p=args.p)
    xs = small.geometry.x.to_numpy()
    small.set_c(M._rows(M.DEFAULT_F[2], xs))
    small.set_z(M._rows(M.DEFAULT_G[2], xs).reshape(-1, order="F"))
    small.solve()
    del small
    backend.synchronize()

    # ---- setup (reported, not in `value`): geometry build + upload, AMG hierarchy, factorisation structures
    t_setup = time.time()
    geo = M.fem2d_mpi(args.L, backend=backend)
    A = M.AMG(geo, p=args.p)
    x = geo.x.to_numpy()
    z0 = M._rows(M.DEFAULT_G[2], x).reshape(-1, order="F")
    c = M._rows(M.DEFAULT_F[2], x)
    A.set_c(c)
    A.set_solver(args.solver)
    A.prepare()                  # operators, Hessian plan, factorisation structures: setup, not solve
    backend.synchronize()
    t_setup = time.time() - t_setup
    n = A.n
    NL = A.level_size(A.L - 1)[0]

    def one_solve():
        A.set_z(z0)
        return A.solve(verbose=args.verbose, solver=args.solver)

    for _ in range(args.warmup):
        one_solve()

    def fence():
        backend.synchronize()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    fence()
    comm0 = backend.comm_stats() if sharded else None
    t0 = time.perf_counter()
    sols = [one_solve() for _ in range(args.steps)]
    backend.synchronize()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    elapsed = max_over_ranks(elapsed, dist, "cpu" if args.rehearse_one_gpu else "cuda")
    if dist is not None:
        dist.barrier()
    comm1 = backend.comm_stats() if sharded else None
    z_final = A.get_z()          # collective on a sharded context: every rank calls it

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
        # (not measured in this run: `traffic_source` names the committed file; null when no committed pass covers the kernel)
        traffic, traffic_source = None, None
        for name in ("r3_pmc_traffic.json", "r2_pmc_traffic.json"):
            try:
                pmc = json.load(open(os.path.join(ROOT, "profiles", name)))
                if pmc.get("workload") == "fem2d L=%d p=%g" % (args.L, args.p) and dom in pmc["kernels"]:
                    traffic = pmc["kernels"][dom]["hbm_bytes_per_launch"]
                    traffic_source = "profiles/" + name + " (committed rocprofv3 --pmc passes of this command, not this run)"
                    break
            except Exception:
                continue
        roofline = dict(bound="hbm", kernel=dom, achieved=achieved, peak=HBM_PEAK_GBS, unit="GB/s",
                        frac=achieved / HBM_PEAK_GBS, traffic=traffic, traffic_source=traffic_source,
                        avg_launch_us=1e3 * kd["ms"] / max(kd["launches"], 1), launches_timed=kd["launches"],
                        algorithmic_bytes_per_launch=kd["bytes"] / max(kd["launches"], 1),
                        note="latency-bound at this size: every working set is L2/Infinity-Cache resident and the "
                             "factorisation + solve is a dependent chain of short launches (DESIGN.md sections 4b, 5)",
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
                       ("one solve, row-block sharded x%d (factorisation split by subtrees that follow the row blocks, top "
                        "replicated; RCCL allreduce of the gradient, of the subtree roots' Schur complements + the Hessian "
                        "entries among separator unknowns, and of the assembled step)" % world
                        if sharded else "replicas x%d (independent solves, not sharded)" % world)},
            "total_solve_s": elapsed / args.steps, "setup_s": t_setup,
            "total_solve_s_incl_setup": t_setup + elapsed / args.steps,
            "newton_steps_per_solve": newton_steps / args.steps,
            "linear_solver": "gpu multifrontal Cholesky (csrc/gpuchol.hip)" if args.solver == "gpu" else
                             "V-cycle-preconditioned CG, matrix-free Hessian (csrc/mg.hip), direct fallback",
            "linear_solve_s_per_solve": sum(s["time_factor"] for s in sols) / args.steps,
            "barrier_spmv_kernel_s_per_solve": sum(est[k] for k in kern if not k.startswith("chol_")) / 1e3 / args.steps,
            "t_final": float(last["ts"][-1]), "c_dot_Dz_final": float(last["c_dot_Dz"][-1]),
            "roofline": roofline,
        }
        # parity of the timed result: z against the CPU oracle's committed vector for this workload (tests/golden, data only)
        gold = os.path.join(ROOT, "tests", "golden", "large_fem2d_L%d_p%s.npz" % (args.L, str(args.p).replace(".", "_")))
        if os.path.exists(gold):
            gz = np.load(gold)
            zo = gz["z"].reshape(-1, order="F")
            relz = float(np.linalg.norm(z_final - zo) / np.linalg.norm(zo))
            relc = float(abs(out["c_dot_Dz_final"] - gz["c_dot_Dz"][-1]) / abs(gz["c_dot_Dz"][-1]))
            PARITY_TOL = 1e-10      # the gate below uses the tolerance it reports
            out["parity"] = {"z_rel_l2_vs_oracle": relz, "c_dot_Dz_rel_vs_oracle": relc, "tolerance": PARITY_TOL,
                             "oracle_newton_steps": int(gz["its"].sum()), "fixture": os.path.basename(gold)}
            if "z_centre" in gz.files:       # the oracle's end point polished to the exact centre (tests/golden/polish_centre.py)
                zc = gz["z_centre"].reshape(-1, order="F")
                out["parity"]["z_rel_l2_vs_oracle_centre"] = float(np.linalg.norm(z_final - zc) / np.linalg.norm(zc))
            if not (relz < PARITY_TOL and relc < 1e-9):
                print(json.dumps(out))
                raise SystemExit("bench.py: the timed solve does not reproduce the oracle's z (rel l2 %.3e)" % relz)
        if args.solver == "pcg":
            out["pcg"] = {k: sum(s["pcg"][k] for s in sols) for k in ("solves", "iterations", "fallbacks", "seconds")}
            out["pcg"]["gave_up_at_newton_system"] = last["pcg"]["gave_up_at"]
        if sharded:
            st = {k: comm1[k] - comm0[k] for k in comm1}      # the timed solves only
            out["communicator"] = ("library-owned RCCL (ncclAllReduce on the context stream)" if backend._allreduce_cb is None
                                   else "torch.distributed callback (host-synchronised)")
            out["multi_gpu_status"] = ("rehearsal on one GPU over gloo: control flow and parity only" if args.rehearse_one_gpu else
                                       "measured on %d GPUs" % world)
            info = A.chol_info()
            out["allreduce"] = dict(st, per_newton_step={"collectives": st["calls"] / max(newton_steps, 1),
                                                         "bytes": st["bytes"] / max(newton_steps, 1)},
                                    hessian_values_stay_on_their_rank=info["values_local"],
                                    factorisation_split_over=info["split_world"])
        if args.probe_L > 0 and world == 1:      # N=1 only, like the CPU baseline
            # secondary evidence for the bandwidth-shaped kernels: the same kernels on the
            # workload mesh (cache resident, launch bound) and on a mesh whose rotating working set exceeds the 256 MiB
            # Infinity Cache
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
