#!/usr/bin/env python3
"""bench.py -- headline benchmark of the PCCSNet modality-clustering hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

Workload (BASELINE.json configs[3], the configuration the metric and the north-star target are
quoted on): k-means over synthetic encodings, N = 10,000,000 points, D = 128, K = 1024, fixed
init.  A "step" is one full Lloyd iteration over all N points: fused fp32-MFMA distance +
argmin (E-step), double-accumulated centroid sums (M-step), [all-reduce of sums/counts over
RCCL when N > 1], averaging / shift / convergence flags.  The N points are sharded over the
ranks (strong scaling: total work fixed).  Convergence is disabled (tol < 0) so that exactly
W + K iterations run; the run aborts if the loop stopped early.

The JSON line also carries
  roofline      for the dominant kernel (E-step): algorithmic 2*N*K*D flop / launch over its
                average duration measured with events on the launch stream, against the fp32
                matrix peak of MI355X_MICROARCH.md (157.3 TFLOP/s);
  cpu_baseline  the reference's CPU path -- scikit-learn's KMeans(init=C0, n_init=1), what
                PCCSNet/utils/utility.py:121 calls -- on a bounded sample of the same workload at the 8
                threads the reference pins (utility.py:8), with the all-cores figure beside it; the
                CPU oracle port (oracle/liboracle.so) is the fallback when sklearn is absent and is
                reported as cpu_baseline_port (rank 0, N = 1 only);
and, as extra keys, the M-step kernel's HBM rate, the second half of the metric -- best-of-K ADE/FDE
trajectories/s on configs[4] (1M agents x K=100 x T=12) with the reference's per-agent torch loop and
NumPy loop as its CPU baselines -- and reference_configs: the whole KMeans(K, n_init=20) fit at the
reference's own sizes (configs[0..2]: the five ETH/UCY splits N=30 128 .. 76 646 / K=200, SDD N=17 970 / K=400) with its dominant
kernel's roofline.
"""
import argparse
import json
import os
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: what RCCL across processes needs on this pool

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "modular-trajectory-prediction-toolkit_amd"))

PEAK_FP32_MATRIX_TFLOPS = 157.3  # MI355X_MICROARCH.md, "Peak FP32 (matrix)"
PEAK_HBM_GBPS = 8000.0           # MI355X_MICROARCH.md, HBM3E peak (spec); a read-only stream measures 6.5-7.2 TB/s (profiles/r03_hbm_read_probe.txt)
TRAFFIC_SOURCE = "profiles/pmc_traffic.json (recorded rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, not measured in this run)"


def recorded_traffic(n, d, k, world, eval_cfg=None):
    """HBM bytes per launch from the committed PMC passes (profiles/pmc_traffic.json, written by
    tools/pmc_summary.py from separate `rocprofv3 --pmc FETCH_SIZE` / `WRITE_SIZE` runs of this very
    command).  Counters cannot be read from inside the process, so the JSON line carries the
    recorded figure, and only when the workload is the one the passes were taken on."""
    try:
        t = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    except (OSError, ValueError):
        return {}
    if world != 1 or (t.get("n"), t.get("d"), t.get("k")) != (n, d, k):
        return {}
    return t.get("hbm_bytes_per_launch", {})


def make_encodings(n, d, k_true, seed, device):
    """X = tanh(M[z] + 0.15*eps), M ~ N(0, 0.5^2) [k_true, d]; values in (-1, 1)."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    gm = torch.Generator(device=device)
    gm.manual_seed(0)  # the mixture is the same on every rank
    M = torch.randn(k_true, d, generator=gm, device=device) * 0.5
    X = torch.empty((n, d), dtype=torch.float32, device=device)
    chunk = 1 << 20
    for s in range(0, n, chunk):
        e = min(n, s + chunk)
        z = torch.randint(0, k_true, (e - s,), generator=g, device=device)
        X[s:e] = torch.tanh(M[z] + 0.15 * torch.randn(e - s, d, generator=g, device=device))
    return X


def make_trajectories(n, k, t, seed, device):
    """gt = cumsum(N(0, 0.3^2)), pred = gt + N(0, 0.2^2)."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    gt = torch.cumsum(torch.randn(n, t, 2, generator=g, device=device) * 0.3, dim=1).contiguous()
    pred = torch.empty((n, k, t, 2), dtype=torch.float32, device=device)
    chunk = 1 << 16
    for s in range(0, n, chunk):
        e = min(n, s + chunk)
        pred[s:e] = gt[s:e, None] + torch.randn(e - s, k, t, 2, generator=g, device=device) * 0.2
    return pred, gt


def cpu_baseline_kmeans(Xs, C0, k, d):
    """One Lloyd iteration (E-step + M-step sums) of the CPU oracle port on a sample."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import kmeans_oracle as O
    cores = int(O.lib().orc_num_threads())
    c2 = O.row_sqnorm(C0)
    t0 = time.perf_counter()
    labels = O.assign(Xs, C0, c2, backend="chain")
    O.accumulate(Xs, labels, k)
    dt = time.perf_counter() - t0
    out = {"value": Xs.shape[0] * k * d / dt, "unit": "distance-evals/s", "cores": cores, "kind": "port",
           "sample": f"1 Lloyd iteration (oracle.c E-step, OpenMP, + M-step sums) on the first {Xs.shape[0]} "
                     f"points of the workload, K={k}, D={d}; {dt:.2f} s",
           "iters_per_s_full_n": None}
    return out, labels


def sklearn_baseline(Xs, C0, k, d, threads):
    """sklearn KMeans(init=C0, n_init=1, max_iter=2) -- the reference's CPU path (PCCSNet/utils/utility.py:121;
    scikit-learn is its third-party dependency) -- under a thread limit (utility.py:8 pins 8)."""
    try:
        import warnings
        from sklearn.cluster import KMeans
        from threadpoolctl import threadpool_limits
    except Exception:  # noqa: BLE001
        return None
    try:
        with warnings.catch_warnings(), threadpool_limits(limits=threads):
            warnings.simplefilter("ignore")
            t0 = time.perf_counter()
            km = KMeans(n_clusters=k, init=C0, n_init=1, max_iter=2, tol=0).fit(Xs)
            dt = time.perf_counter() - t0
    except Exception as exc:  # noqa: BLE001  (the oracle port then stands in as the CPU baseline)
        print(f"bench.py: sklearn baseline at {threads} threads failed: {exc!r}", file=sys.stderr)
        return None
    # 2 Lloyd iterations + the final E-step = 3 distance passes
    return {"value": 3 * Xs.shape[0] * k * d / dt, "unit": "distance-evals/s", "cores": threads, "kind": "reference",
            "sample": f"sklearn KMeans(init=C0, n_init=1, max_iter=2) on the first {Xs.shape[0]} points of the workload, "
                      f"{threads} threads; {dt:.2f} s ({km.n_iter_} iterations + final E-step = 3 distance passes)"}


def host_cores():
    """This job's share of the host: the GPU box exports OMP_NUM_THREADS = its per-GPU share (16) while
    os.cpu_count() shows the whole machine."""
    try:
        c = len(os.sched_getaffinity(0))
    except AttributeError:
        c = os.cpu_count() or 1
    env = os.environ.get("OMP_NUM_THREADS", "")
    return min(c, int(env)) if env.isdigit() and int(env) > 0 else c


def ms_of(fn, reps=3):
    """median wall time of fn() in ms, device synchronised on both sides"""
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    return float(np.median(ts))


def reference_config(name, n, k, n_val, seed, device, with_cpu):
    """BASELINE configs[0..2]: the reference's own call, KMeans(K, n_init=20).fit on [n, 96] encodings, then
    get_modality and the best-of-20 evaluation, at the reference's sizes (SURVEY.md 8(d) inputs 1-3)."""
    from pccs_accel import clustering, seeding
    from pccs_accel import kernels as K
    d, n_init = 96, 20
    X = make_encodings(n, d, k, seed=seed, device=device)
    fit = {}

    def run_fit():
        np.random.seed(seed)
        fit["r"] = clustering.kmeans_fit(X, k, n_init=n_init)

    run_fit()  # warm-up (kernel images, allocator)
    fit_ms = ms_of(run_fit)
    # the two halves of the fit, separately
    Xc = X.clone()
    K.sub_rowvec_(Xc, fit["r"]["x_mean"])
    rs = np.random.RandomState(seed)
    draws = [seeding.draw(rs, n, k) for _ in range(n_init)]
    seeds = {}

    def run_seed():
        seeds["c"], _ = seeding.kmeans_plusplus_batched(Xc, k, draws)

    run_seed()
    seed_ms = ms_of(run_seed)
    lloyd_ms = ms_of(lambda: K.lloyd_batched(Xc, seeds["c"], tol_abs=fit["r"]["tol_abs"], check_every=4))
    # dominant kernel: the k-means++ candidate-distance GEMM (fp64 on the matrix pipe), timed through the
    # step-level entry that launches the same kernel on the same shapes (20 restarts x L candidates x n points)
    l = 2 + int(np.log(k))
    ops = seeding.DeviceSeedOps()
    yy = ops.row_sqnorm(Xc)
    cand = Xc[torch.randint(0, n, (n_init * l,), device=device)].view(n_init, l, d).contiguous()
    closest = torch.rand((n_init, n), device=device) * 4
    ops.candidate_dists(Xc, yy, cand, closest)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 50
    e0.record()
    for _ in range(reps):
        ops.candidate_dists(Xc, yy, cand, closest)
    e1.record()
    torch.cuda.synchronize()
    step_ms = e0.elapsed_time(e1) / reps
    flop = 2.0 * n * d * n_init * l   # per launch; the kernel computes whole 8-row tiles: 8 / l times as many
    # fp64 peak: 78.6 TFLOP/s is AMD's data-sheet figure for MI355X (vector = matrix); MI355X_MICROARCH.md lists no
    # fp64 row.  Measured on the box (profiles/r01_mfma_f64_rate.txt): v_mfma_f64_16x16x4 47.8, v_fma_f64 69.5.
    PEAK_F64_MATRIX = 78.6
    out = {"workload": f"{name}: KMeans(K={k}, n_init={n_init}).fit on N={n} x D={d} synthetic encodings + get_modality "
                       f"+ best-of-20 ADE/FDE on {n_val} agents",
           "fit_ms": fit_ms, "seeding_ms": seed_ms, "lloyd_ms": lloyd_ms, "n_iter_best": int(fit["r"]["n_iter"]),
           "inertia": float(fit["r"]["inertia"]),
           "roofline": {"bound": "mfma", "kernel": "k_kpp_dist_mfma (+ staging and potential sums of one seeding step)",
                        "achieved": flop / (step_ms * 1e-3) / 1e12, "peak": PEAK_F64_MATRIX, "unit": "TFLOP/s (fp64)",
                        "frac": flop / (step_ms * 1e-3) / 1e12 / PEAK_F64_MATRIX, "traffic": None,
                        "ms_per_launch": step_ms, "algorithmic_flop_per_launch": flop,
                        "launches_per_fit": k - 1}}
    labels = fit["r"]["labels"]
    out["get_modality_ms"] = ms_of(lambda: K.cluster_means(*K.centroid_sums(X, labels, k)))
    pred, gt = make_trajectories(n_val, 20, 12, seed=seed + 1, device=device)
    out["best_of_20_ms"] = ms_of(lambda: K.best_of_k(pred, gt, mode=0))
    if with_cpu:
        try:
            import warnings
            from sklearn.cluster import KMeans
            from threadpoolctl import threadpool_limits
            Xh = X.cpu().numpy()
            with warnings.catch_warnings(), threadpool_limits(limits=8):
                warnings.simplefilter("ignore")
                np.random.seed(seed)
                t0 = time.perf_counter()
                km = KMeans(n_clusters=k, n_init=2).fit(Xh)
                cdt = time.perf_counter() - t0
            out["cpu_baseline"] = {"value": cdt * 1e3 * n_init / 2, "unit": "ms per fit", "cores": 8, "kind": "reference",
                                   "sample": f"sklearn KMeans(K, n_init=2) on the same points, 8 threads "
                                             f"(PCCSNet/utils/utility.py:8): {cdt:.2f} s, scaled to n_init=20",
                                   "inertia_best_of_2": float(km.inertia_)}
        except Exception as exc:  # noqa: BLE001
            out["cpu_baseline"] = {"error": repr(exc)}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--n", "--points", dest="n", type=int, default=10_000_000)
    ap.add_argument("--d", type=int, default=128)
    ap.add_argument("--k", type=int, default=1024)
    ap.add_argument("--eval-n", type=int, default=1_000_000)
    ap.add_argument("--eval-k", type=int, default=100)
    ap.add_argument("--eval-t", type=int, default=12)
    ap.add_argument("--cpu-sample", type=int, default=10_000_000)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-eval", action="store_true")
    ap.add_argument("--no-ref-configs", action="store_true", help="skip the reference-sized configs (the five ETH/UCY splits, SDD)")
    ap.add_argument("--no-convergence", action="store_true",
                    help="skip the run-to-convergence fit (profiled runs: the per-kernel averages are then those of the timed steps)")
    ap.add_argument("--loop", choices=["library", "python"], default="library",
                    help="library: the K timed iterations are ONE call of mtp_lloyd_sharded_f32 (loop, RCCL all-reduce and "
                         "event timing inside libmtp_accel.so); python: the step-level driver ShardedLloyd.run_fixed "
                         "(torch.distributed all-reduce, flags read one step late) -- always used under --backend gloo")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo for rehearsals)")
    ap.add_argument("--one-device", action="store_true",
                    help="rehearsal only: every rank uses cuda:0 (needs --backend gloo); not a valid scaling result")
    ap.add_argument("--force-collectives", action="store_true",
                    help="rehearsal only: with one rank (launched through torch.distributed.run), create the RCCL "
                         "communicator anyway and issue every collective of the N > 1 path")
    args = ap.parse_args()

    # `python bench.py --gpus N` with N > 1 and no launcher around it: start the N ranks ourselves.  The parent has
    # not touched the GPU (importing torch does not); the ranks are fresh child processes of torch.distributed.run
    # (never an exec of this process), rank 0's JSON line reaches stdout through the inherited descriptor, and the
    # parent exits with the launcher's return code.
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        import socket
        import subprocess
        with socket.socket() as sock:
            sock.bind(("127.0.0.1", 0))
            port = sock.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        print("bench.py: launching", " ".join(cmd), file=sys.stderr)
        sys.exit(subprocess.run(cmd).returncode)

    # stdout carries ONE JSON line.  RCCL prints a version banner to the process's stdout when a communicator is
    # created: everything else written to fd 1 during the run goes to stderr, the JSON line to the real stdout.
    real_stdout = os.fdopen(os.dup(1), "w")
    sys.stdout.flush()
    os.dup2(2, 1)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if args.gpus != 1 or world != 1:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch with torch.distributed.run",
                  file=sys.stderr)
            sys.exit(2)
    if args.one_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    collective = world > 1 or args.force_collectives
    if collective:
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=device)
        else:
            dist.init_process_group(backend=args.backend)

    from pccs_accel import dist as pdist
    from pccs_accel import kernels as K

    def barrier():
        if collective:
            dist.barrier()
        torch.cuda.synchronize()

    def allreduce(t, op=dist.ReduceOp.SUM):
        """RCCL on the device buffer; under gloo (rehearsals) staged through the host."""
        if args.backend == "gloo":
            h = t.cpu()
            dist.all_reduce(h, op=op)
            t.copy_(h)
        else:
            dist.all_reduce(t, op=op)

    n, d, k = args.n, args.d, args.k
    lo, hi = pdist.shard_bounds(n, rank, world)
    n_local = hi - lo
    X = make_encodings(n_local, d, k, seed=1000 + rank, device=device)
    # fixed init: K rows of rank 0's shard, the same on every rank
    perm = torch.randperm(min(n_local, 1 << 22), generator=torch.Generator().manual_seed(3))[:k].to(device)
    C0 = X[perm].contiguous()
    if collective:  # rank 0's rows everywhere (a sum with zeros: one code path for RCCL and gloo)
        if rank != 0:
            C0.zero_()
        allreduce(C0)
    # centring as KMeans.fit does (global mean)
    col = K.col_sums(X)
    if collective:
        allreduce(col)
    mean32 = (col / n).to(torch.float32)
    K.sub_rowvec_(X, mean32)
    K.sub_rowvec_(C0, mean32)

    TOL_NEVER = -1.0            # shift_tot <= tol never holds: no early exit inside the timed region
    MAX_IT = 1 << 30
    total_steps = args.warmup + args.steps
    # MTP_RCCL_LIB names another collective library for libmtp_accel.so to bind (tests/stub_nccl: shared memory, several
    # processes on one GPU): with it the in-library loop can be rehearsed on one device, which RCCL itself refuses
    stub_lib = os.environ.get("MTP_RCCL_LIB", "")
    use_library = args.loop == "library" and ((args.backend != "gloo" and not args.one_device) or bool(stub_lib))
    loop_note = "library (mtp_lloyd_sharded_f32)"
    comm = None
    if use_library and collective:
        try:
            comm = pdist.RcclComm(device)
        except Exception as exc:  # noqa: BLE001  (librccl missing / communicator refused: the step-level driver still works)
            use_library = False
            loop_note = f"python (library communicator failed: {exc!r})"
        # every rank must take the same path
        ok = torch.tensor([1 if use_library else 0], dtype=torch.int32, device=device)
        allreduce(ok, dist.ReduceOp.MIN)
        if int(ok.item()) == 0 and use_library:
            use_library = False
            comm.close()
            comm = None
            loop_note = "python (another rank could not create the library communicator)"

    if use_library:
        # Warm-up call: W iterations from C0 (this workload's empty-cluster relocations happen here), then the timed
        # call: K iterations from the warmed-up centres as ONE library call -- per iteration E-step, local sums,
        # in-stream ncclAllReduce of the packed sums (N > 1), M-step finish; the host is out of the loop until the
        # single flag check after iteration K.  No closing E-step / inertia inside the timed region
        # (MTP_LLOYD_NO_FINAL_ESTEP): exactly K steps.
        n_relocated = None
        _, _, C_w, it_w = pdist.lloyd_sharded_native(comm, X, C0, lo, n, max_iter=max(args.warmup, 1), tol_abs=TOL_NEVER,
                                                     check_every=1, final_estep=False)
        barrier()
        t0 = time.perf_counter()
        _, _, C_t, it_t, phase = pdist.lloyd_sharded_native(comm, X, C_w, lo, n, max_iter=args.steps, tol_abs=TOL_NEVER,
                                                            check_every=args.steps, want_phase_ms=True, final_estep=False)
        barrier()
        dt = time.perf_counter() - t0
        if collective:
            tmax = torch.tensor([dt], dtype=torch.float64, device=device)
            allreduce(tmax, dist.ReduceOp.MAX)
            dt = float(tmax.item())
        if it_t != args.steps:
            print(f"bench.py: the library loop ran {it_t} iterations, wanted {args.steps}; timing invalid", file=sys.stderr)
            sys.exit(3)
        ms_assign, ms_accum, ms_allreduce, ms_mstep = phase
        ms_rest = ms_allreduce + ms_mstep
        noop = []
        sl = None
        # SURVEY 8(d) item 4, "run-to-convergence count": the same points from the same C0 with sklearn's tolerance
        # (tol = 1e-4 * mean column variance, _kmeans.py:279-287) to the end, outside the timed region
        convergence = lab_c = None
        if not args.no_convergence:
            sq = K.col_sqdev(X, torch.zeros(d, dtype=torch.float64, device=device))  # X is centred: deviations from 0
            if collective:
                allreduce(sq)
            tol_abs = float((sq / n).mean().item()) * 1e-4
            barrier()
            t0 = time.perf_counter()
            lab_c, inertia_c, C_c, it_c = pdist.lloyd_sharded_native(comm, X, C0, lo, n, max_iter=300, tol_abs=tol_abs, check_every=8)
            barrier()
            convergence = {"n_iter": int(it_c), "total_ms": (time.perf_counter() - t0) * 1e3, "inertia": float(inertia_c),
                           "tol_abs": tol_abs, "max_iter": 300,
                           "note": "KMeans(init=C0, n_init=1, tol=1e-4) from the bench's fixed init on all N points, incl. the "
                                   "closing E-step and the inertia; flags checked every 8 iterations"}
        if comm is not None:
            comm.close()
    else:
        if args.loop == "library" and loop_note.startswith("library"):
            loop_note = "python (gloo / one-device rehearsal)"
        elif args.loop == "python":
            loop_note = "python (ShardedLloyd.run_fixed)"
        sl = pdist.ShardedLloyd(X, C0, row_offset=lo, force_collectives=args.force_collectives)
        # The loop state (an empty cluster to relocate, as sklearn does; they do occur in the first
        # iterations of this workload) is read one step late from a pinned snapshot, so the device
        # never waits for the host (ShardedLloyd.run_fixed).  An attempt enqueued behind a paused
        # iteration is a no-op and is enqueued again; it stays inside the timed region but is left
        # out of the per-kernel averages.
        # The warm-up steps (where this workload's relocations happen) check the flags synchronously, so that
        # no no-op launches blur the per-kernel averages of a profiled run.
        n_relocated = 0
        for _ in range(args.warmup):
            sl.assign()
            sl.accumulate()
            sl.reduce()
            sl.mstep(TOL_NEVER, MAX_IT)
            n_relocated += sl.finish_step(TOL_NEVER, MAX_IT)

        ev = {}

        def on_phase(attempt, phase):
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            ev.setdefault(attempt, [None] * 4)[phase] = e

        barrier()
        t0 = time.perf_counter()
        n_reloc_timed, noop, attempts, _ = sl.run_fixed(args.steps, TOL_NEVER, MAX_IT, on_phase=on_phase)
        barrier()
        dt = time.perf_counter() - t0
        n_relocated += n_reloc_timed
        if collective:
            tmax = torch.tensor([dt], dtype=torch.float64, device=device)
            allreduce(tmax, dist.ReduceOp.MAX)
            dt = float(tmax.item())
        ev = [ev[a] for a in range(attempts) if a not in set(noop)]

        flags = sl.read_flags()
        if int(flags[1]) != total_steps or int(flags[0]) != 0:
            print(f"bench.py: Lloyd loop stopped early (state={flags[0]}, n_iter={flags[1]}, wanted {total_steps}); "
                  "timing invalid", file=sys.stderr)
            sys.exit(3)

        ms_assign = float(np.mean([e[0].elapsed_time(e[1]) for e in ev]))
        ms_accum = float(np.mean([e[1].elapsed_time(e[2]) for e in ev]))
        ms_rest = float(np.mean([e[2].elapsed_time(e[3]) for e in ev]))
        ms_allreduce = ms_mstep = None
        convergence = lab_c = None
    ms_step = dt / args.steps * 1e3
    evals_per_s = float(n) * k * d * args.steps / dt

    traffic = recorded_traffic(n, d, k, world)
    # roofline of the dominant kernel, per launch, on this rank's shard
    flop = 2.0 * n_local * k * d
    achieved_tf = flop / (ms_assign * 1e-3) / 1e12
    roofline = {"bound": "mfma", "kernel": "k_assign_mfma", "achieved": achieved_tf, "peak": PEAK_FP32_MATRIX_TFLOPS,
                "unit": "TFLOP/s", "frac": achieved_tf / PEAK_FP32_MATRIX_TFLOPS, "traffic": traffic.get("k_assign_mfma"),
                "traffic_source": TRAFFIC_SOURCE if traffic.get("k_assign_mfma") is not None else None,
                "ms_per_launch": ms_assign, "algorithmic_flop_per_launch": flop}
    acc_bytes = n_local * d * 4.0 + n_local * 4.0 + k * d * 8.0 + k * 8.0
    roofline_update = {"bound": "hbm", "kernel": "k_accumulate", "achieved": acc_bytes / (ms_accum * 1e-3) / 1e9,
                       "peak": PEAK_HBM_GBPS, "unit": "GB/s", "frac": acc_bytes / (ms_accum * 1e-3) / 1e9 / PEAK_HBM_GBPS,
                       "traffic": traffic.get("k_accumulate"),
                       "traffic_source": TRAFFIC_SOURCE if traffic.get("k_accumulate") is not None else None,
                       "ms_per_launch": ms_accum, "algorithmic_bytes_per_launch": acc_bytes}

    out = {
        "metric": "k-means iters/sec (N×K×D distance evals/sec) + best-of-K ADE/FDE traj/sec",
        "value": evals_per_s, "unit": "distance-evals/s (N*K*D MACs of the cross term per second, whole job)",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_step,
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic" + (" (REHEARSAL: all ranks on one device, not a scaling result)" if args.one_device else "")
                + (" (REHEARSAL: one-rank RCCL communicator, collectives forced)" if args.force_collectives and world == 1 else ""),
        "config": {"workload": f"k-means Lloyd iterations, synthetic encodings N={n} D={d} K={k}, fixed init, "
                               f"points sharded over {world} GPU(s)" + (", RCCL all-reduce of sums/counts" if collective else ""),
                   "n": n, "d": d, "k": k, "parallelism": f"points/{world}"},
        "iters_per_s": args.steps / dt, "loop": loop_note, "clusters_relocated": n_relocated,
        "noop_attempts_in_timed_region": len(noop),
        "ms_breakdown": {"assign": ms_assign, "accumulate": ms_accum, "allreduce+mstep": ms_rest,
                         "allreduce": ms_allreduce, "mstep_finish": ms_mstep,
                         "allreduce_pct_of_step": (100.0 * ms_allreduce / ms_step) if ms_allreduce is not None else None},
        "convergence": convergence,
        "roofline": roofline, "roofline_update": roofline_update,
    }

    # ---- best-of-K ADE/FDE (second half of the metric), agents sharded, no exchange ----
    if not args.no_eval:
        sl = None
        elo, ehi = pdist.shard_bounds(args.eval_n, rank, world)
        pred, gt = make_trajectories(ehi - elo, args.eval_k, args.eval_t, seed=4 + rank, device=device)
        reps = 30
        for _ in range(5):  # the first launches after a 9.6 GB allocation include one-off slow passes
            K.best_of_k(pred, gt, mode=0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        barrier()
        t0 = time.perf_counter()
        e0.record()
        for i in range(reps - 1):
            K.best_of_k(pred, gt, mode=i % 3)
        e1.record()
        # the last pass goes through the sharded entry: local scoring + the one all-reduce of the two sums
        r = pdist.best_of_k_sharded(pred, gt, args.eval_n, mode=(reps - 1) % 3, force_collectives=args.force_collectives)
        barrier()
        edt = time.perf_counter() - t0
        if collective:
            tmax = torch.tensor([edt], dtype=torch.float64, device=device)
            allreduce(tmax, dist.ReduceOp.MAX)
            edt = float(tmax.item())
        reps_ev = reps - 1
        ms_eval = e0.elapsed_time(e1) / reps_ev
        ebytes = (ehi - elo) * (args.eval_k * args.eval_t * 2 * 4.0 + args.eval_t * 2 * 4.0 + 16.0)
        out["best_of_k"] = {
            "value": float(args.eval_n) * args.eval_k * reps / edt, "unit": "traj/s (candidate trajectories scored, whole job)",
            "config": {"workload": f"best-of-K ADE/FDE, {args.eval_n} agents x K={args.eval_k} x T={args.eval_t}, modes 0/1/2"},
            "ms_per_pass": edt / reps * 1e3, "mean_ade_fde_last_pass": [r["mean_ade"], r["mean_fde"]],
            "roofline": {"bound": "hbm", "kernel": "k_best_of_k", "achieved": ebytes / (ms_eval * 1e-3) / 1e9,
                         "peak": PEAK_HBM_GBPS, "unit": "GB/s", "frac": ebytes / (ms_eval * 1e-3) / 1e9 / PEAK_HBM_GBPS,
                         "traffic": traffic.get("k_best_of_k") if (args.eval_n, args.eval_k, args.eval_t) == (1_000_000, 100, 12) else None,
                         "traffic_source": TRAFFIC_SOURCE if traffic.get("k_best_of_k") is not None else None,
                         "ms_per_launch": ms_eval, "algorithmic_bytes_per_launch": ebytes}}
        # CPU baselines of this half beside it (rank 0, N = 1 only), on bounded samples of the same trajectories:
        # the reference's per-agent torch loop (PCCSNet/train_steps.py:214-232) and the NumPy loop of
        # StimulusVerification/verify_predictions.py:238-248 (float64, best of the first 20), both restated in
        # oracle/glue_oracle.py; the oracle's scalar C port is what the parity gate compares with
        if world == 1 and not args.no_cpu:
            try:
                sys.path.insert(0, os.path.join(ROOT, "oracle"))
                import glue_oracle as G
                import kmeans_oracle as O
                nt = min(20_000, ehi - elo)
                pt, gtt = pred[:nt].cpu(), gt[:nt].cpu()
                torch_threads = torch.get_num_threads()
                t0 = time.perf_counter()
                tv = G.evaluate_loop_torch(pt, gtt, mode=0)
                tdt = time.perf_counter() - t0
                out["best_of_k"]["cpu_baseline"] = {
                    "value": nt * float(args.eval_k) / tdt, "unit": "traj/s", "cores": torch_threads, "kind": "port",
                    "sample": f"the reference's evaluate loop (train_steps.py:214-232: one agent at a time, torch CPU ops, "
                              f".item() per agent) on the first {nt} agents, mode 0; {tdt:.2f} s",
                    "mean_ade_fde": [float(tv[0]), float(tv[1])]}
                nv = min(100_000, ehi - elo)
                p64, g64 = pred[:nv, :20].cpu().numpy().astype(np.float64), gt[:nv].cpu().numpy().astype(np.float64)
                t0 = time.perf_counter()
                G.verify_best_of_numpy(p64, g64, best_of=20)
                vdt = time.perf_counter() - t0
                out["best_of_k"]["cpu_baseline_numpy"] = {
                    "value": nv * 20.0 / vdt, "unit": "traj/s", "cores": 1, "kind": "port",
                    "sample": f"verify_predictions.py:238-248 (np.linalg.norm + Python loop over agents, float64) on the first "
                              f"{nv} agents x best-of-20; {vdt:.2f} s"}
                del p64, g64
                ns = min(1_000_000, ehi - elo)
                ph, gh = pred[:ns].cpu().numpy(), gt[:ns].cpu().numpy()
                t0 = time.perf_counter()
                o = O.best_of_k(ph, gh, mode=0)
                cdt = time.perf_counter() - t0
                g = K.best_of_k(pred[:ns], gt[:ns], mode=0)
                out["best_of_k"]["cpu_baseline_port"] = {
                    "value": ns * float(args.eval_k) / cdt, "unit": "traj/s", "cores": 1, "kind": "port",
                    "sample": f"oracle.c best-of-K (scalar, one thread) on the first {ns} agents, mode 0; {cdt:.2f} s"}
                out["best_of_k"]["parity"] = {
                    "sample_agents": int(ns), "argmin_mismatches_vs_cpu_oracle": int((g["idx"].cpu().numpy() != o["idx"]).sum()),
                    "max_rel_err_min_ade": float(np.max(np.abs(g["min_ade"].cpu().numpy() - o["min_ade"]) / np.maximum(o["min_ade"], 1e-30))),
                    "torch_loop_mean_ade_rel_err": float(abs(tv[0] - float(K.best_of_k(pred[:nt], gt[:nt], mode=0)["sums"][0].item()) / nt) / tv[0])}
            except Exception as exc:  # noqa: BLE001  (a CPU leg that fails is reported, the GPU numbers stay)
                out["best_of_k"]["cpu_baseline_error"] = repr(exc)
                print(f"bench.py: best-of-K CPU legs failed: {exc!r}", file=sys.stderr)
        del pred, gt

    # ---- CPU baseline on a bounded sample (rank 0, N = 1 only) ----
    if world == 1 and not args.no_cpu:
        ns = min(args.cpu_sample, n_local)
        Xs = X[:ns].cpu().numpy()
        C0h = C0.cpu().numpy()
        port, cpu_labels = cpu_baseline_kmeans(Xs, C0h, k, d)
        port["iters_per_s_full_n"] = port["value"] / (float(n) * k * d)
        # parity gate run with the benchmark: the GPU E-step on the same sample and centres
        gl = K.assign(X[:ns].contiguous(), C0).cpu().numpy()
        out["parity"] = {"sample_points": int(ns), "label_mismatches_vs_cpu_oracle": int((gl != cpu_labels).sum())}
        # centroid gate ("centroids vs get_modality <= 1e-5"): the modality means of the converged
        # labels -- mtp_centroid_sums_f32 + mtp_cluster_means_f32 against the oracle's restatement of
        # PCCSNet/utils/utility.py:98-105 on the same points and labels
        if lab_c is not None:
            import kmeans_oracle as O
            lab_h = lab_c[:ns].cpu().numpy()
            if len(np.unique(lab_h)) == k:
                gm = K.cluster_means(*K.centroid_sums(X[:ns].contiguous(), lab_c[:ns].contiguous(), k))[0].cpu().numpy()
                om = O.get_modality(Xs, lab_h, k)
                scale = np.maximum(np.abs(om), 1e-3 * float(np.abs(om).max()))
                out["parity"]["get_modality_max_rel_err_vs_cpu_oracle"] = float(np.max(np.abs(gm - om) / scale))
                out["parity"]["get_modality_bar"] = 1e-5
            else:
                out["parity"]["get_modality_max_rel_err_vs_cpu_oracle"] = None  # a cluster without sample points
        cores = host_cores()
        sk8 = sklearn_baseline(Xs, C0h, k, d, min(8, cores))
        if sk8 is not None:
            sk8["iters_per_s_full_n"] = sk8["value"] / (float(n) * k * d)
            out["cpu_baseline"] = sk8
            if cores > 8:
                sk_all = sklearn_baseline(Xs, C0h, k, d, cores)
                if sk_all is not None:
                    out["cpu_baseline_all_cores"] = sk_all
            out["cpu_baseline_port"] = port
        else:
            out["cpu_baseline"] = port
    else:
        out["cpu_baseline"] = None

    # ---- the reference's own sizes: whole n_init=20 fits (rank 0, N = 1 only) ----
    if world == 1 and not args.no_ref_configs:
        del X
        torch.cuda.empty_cache()
        # BASELINE configs 0-2: the five ETH/UCY leave-one-out splits (train / val sizes of SURVEY.md 8) and SDD
        out["reference_configs"] = {}
        for key, cfg in (("eth", ("eth split", 73_568, 200, 2_614, 10)), ("hotel", ("hotel split", 76_646, 200, 1_075, 11)),
                         ("univ", ("univ split", 30_128, 200, 24_334, 12)), ("zara1", ("zara1 split", 74_084, 200, 2_356, 13)),
                         ("zara2", ("zara2 split", 66_976, 200, 5_910, 14)), ("sdd", ("SDD", 17_970, 400, 2_829, 20))):
            try:
                out["reference_configs"][key] = reference_config(*cfg, device, not args.no_cpu)
            except Exception as exc:  # noqa: BLE001  (a failed extra must not take the headline line with it; it is reported)
                out["reference_configs"][key] = {"error": repr(exc)}
                print(f"bench.py: reference config {key} failed: {exc!r}", file=sys.stderr)

    if rank == 0:
        real_stdout.write(json.dumps(out) + "\n")
        real_stdout.flush()
    if collective:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
