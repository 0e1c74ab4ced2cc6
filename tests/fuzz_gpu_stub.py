#!/usr/bin/env python3
"""Randomised sweep of the in-library sharded Lloyd loop (mtp_lloyd_sharded_f32) over the stub collectives:
2-8 ranks as processes on cuda:0 (tests/stub_nccl, see tests/test_gpu_dist_stub.py), random shapes, random
and ragged row splits (ranks without rows anywhere), unreachable centres (relocation through the all-gathers),
random check_every, against the one-GPU mtp_lloyd_f32 run of the same case.  Development aid, run on the MI355X box:

    python tests/fuzz_gpu_stub.py [seconds] [seed] [loop|fit]

`fit` sweeps the whole row-sharded KMeans(n_clusters, n_init).fit instead (pccs_accel.dist.kmeans_fit_sharded over
gloo: column statistics handed from rank to rank, k-means++ in lock step with double or reference-order float32
potentials, every restart's Lloyd run, sklearn's keep-the-best rule) against the one-GPU clustering.kmeans_fit with
the same random_state.

Not collected by pytest.  Bars: labels and n_iter identical; centres identical on every rank and equal to the
one-GPU ones to 1e-6 (double sums added in another order); inertia to 1e-9."""
import os
import sys
import tempfile
import time

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG_DIR = os.path.join(ROOT, "modular-trajectory-prediction-toolkit_amd")
STUB = os.path.join(ROOT, "tests", "stub_nccl", "libstubnccl.so")
CASES_PER_SPAWN = 10


def make_case(seed):
    """(X centred, C0, tol, check_every, max_iter); the same on every rank and in the parent."""
    sys.path[:0] = [os.path.join(ROOT, "tests", "golden")]
    import datagen
    rs = np.random.RandomState(seed)
    n, d, k = int(rs.randint(30, 20000)), int(rs.choice([64, 96, 128, 96, 12, 40])), int(rs.randint(1, 120))
    k = min(k, n)
    X = datagen.make_encodings(int(rs.randint(1 << 30)), n, d, max(1, k // 2), 1500, 700)
    X = X - X.mean(axis=0, dtype=np.float64).astype(np.float32)
    C0 = X[rs.choice(n, size=k, replace=False)].copy()
    if k >= 2 and rs.rand() < 0.6:  # unreachable centres: empty clusters, relocation
        C0[rs.choice(k, size=int(rs.randint(1, k)), replace=False)] += np.float32(50.0)
    tol = float(np.mean(np.var(X, axis=0)) * 1e-4) if rs.rand() < 0.8 else 0.0
    return X, C0, tol, int(rs.choice([1, 2, 4, 9, 1000])), int(rs.choice([40, 7, 1]))


def make_bounds(seed, n, world):
    rs = np.random.RandomState(seed ^ 0x5bd1e995)
    kind = rs.randint(3)
    if kind == 0:   # even
        cut = [n * r // world for r in range(world + 1)]
    elif kind == 1:  # ragged
        cut = [0] + sorted(int(c) for c in rs.randint(0, n + 1, size=world - 1)) + [n]
    else:           # at least one rank without rows
        inner = sorted(int(c) for c in rs.randint(0, n + 1, size=world - 2))
        pool = [0] + inner + [n]
        cut = [0] + sorted(inner + [pool[int(rs.randint(len(pool)))]]) + [n]
    return [(cut[r], cut[r + 1]) for r in range(world)]


def worker(rank, world, rdv, seeds, out_dir):
    import faulthandler
    faulthandler.dump_traceback_later(170, exit=True)
    os.environ["MTP_RCCL_LIB"] = STUB
    sys.path[:0] = [PKG_DIR]
    torch.cuda.set_device(0)
    devc = torch.device("cuda:0")
    dist.init_process_group("gloo", init_method=f"file://{rdv}", rank=rank, world_size=world)
    from pccs_accel import dist as pdist
    comm = None
    for i, seed in enumerate(seeds):
        X, C0, tol, ce, max_iter = make_case(seed)
        lo, hi = make_bounds(seed, X.shape[0], world)[rank]
        if comm is None or i % 3 == 0:  # a communicator serves several runs, and is re-made now and then
            if comm is not None:
                comm.close()
            comm = pdist.RcclComm(devc)
        labels, inertia, centers, n_iter = pdist.lloyd_sharded_native(
            comm, torch.from_numpy(np.ascontiguousarray(X[lo:hi])).to(devc), torch.from_numpy(C0).to(devc), lo,
            X.shape[0], max_iter=max_iter, tol_abs=tol, check_every=ce)
        np.savez(os.path.join(out_dir, f"c{seed}_r{rank}.npz"), labels=labels.cpu().numpy(),
                 centers=centers.cpu().numpy(), inertia=inertia, n_iter=n_iter)
    comm.close()
    dist.destroy_process_group()
    faulthandler.cancel_dump_traceback_later()


def make_fit_case(seed):
    sys.path[:0] = [os.path.join(ROOT, "tests", "golden")]
    import datagen
    rs = np.random.RandomState(seed)
    order = str(rs.choice(["f64", "openblas"]))
    n = int(rs.randint(3 * 4096 + 1, 40000)) if order == "openblas" else int(rs.randint(300, 20000))
    d, k = int(rs.choice([64, 96, 128, 16, 48, 80])), int(rs.randint(2, 60))  # the sharded seeding takes D % 16 == 0
    X = datagen.make_encodings(int(rs.randint(1 << 30)), n, d, max(1, k // 2), 1500, 700)
    return np.ascontiguousarray(X), k, int(rs.choice([1, 2, 5, 20])), int(rs.randint(1 << 30)), order


def fit_worker(rank, world, rdv, seeds, out_dir):
    import faulthandler
    faulthandler.dump_traceback_later(170, exit=True)
    sys.path[:0] = [PKG_DIR]
    torch.cuda.set_device(0)
    devc = torch.device("cuda:0")
    dist.init_process_group("gloo", init_method=f"file://{rdv}", rank=rank, world_size=world)
    from pccs_accel import dist as pdist
    from pccs_accel import seeding
    for seed in seeds:
        X, k, n_init, state, order = make_fit_case(seed)
        n = X.shape[0]
        lo, hi = pdist.shard_bounds(n, rank, world, align=seeding.POT_ALIGN if order == "openblas" else 1)
        fit = pdist.kmeans_fit_sharded(torch.from_numpy(X[lo:hi].copy()).to(devc), n, row_offset=lo, n_clusters=k,
                                       n_init=n_init, random_state=state, force_collectives=True, pot_order=order)
        np.savez(os.path.join(out_dir, f"c{seed}_r{rank}.npz"), labels=fit["labels"].cpu().numpy(),
                 centers=fit["centers"].cpu().numpy(), inertia=fit["inertia"], n_iter=fit["n_iter"])
    dist.destroy_process_group()
    faulthandler.cancel_dump_traceback_later()


def main_fit(budget, seed):
    sys.path[:0] = [PKG_DIR]
    from pccs_accel import clustering
    rs = np.random.RandomState(seed)
    devc = torch.device("cuda:0")
    t0 = time.time()
    n_cases = n_checks = 0
    fails = []
    while time.time() - t0 < budget:
        world = int(rs.choice([2, 3, 4]))  # 4 x 4096-row blocks fit every openblas-order case (n > 3 * 4096)
        seeds = [int(s) for s in rs.randint(1 << 30, size=4)]
        fd, rdv = tempfile.mkstemp(prefix="mtp_rdv_")
        os.close(fd)
        with tempfile.TemporaryDirectory() as out_dir:
            mp.spawn(fit_worker, args=(world, rdv, seeds, out_dir), nprocs=world, join=True)
            for s in seeds:
                X, k, n_init, state, order = make_fit_case(s)
                single = clustering.kmeans_fit(torch.from_numpy(X).to(devc), k, n_init=n_init, random_state=state, pot_order=order)
                parts = [np.load(os.path.join(out_dir, f"c{s}_r{r}.npz")) for r in range(world)]
                info = dict(seed=s, world=world, n=X.shape[0], d=X.shape[1], k=k, n_init=n_init, order=order)
                checks = {
                    "fit.n_iter": all(int(p["n_iter"]) == single["n_iter"] for p in parts),
                    "fit.replicated": all(np.array_equal(p["centers"], parts[0]["centers"]) for p in parts),
                    "fit.labels": np.array_equal(np.concatenate([p["labels"] for p in parts]), single["labels"].cpu().numpy()),
                    "fit.centers": np.allclose(parts[0]["centers"], single["centers"].cpu().numpy(), rtol=1e-6, atol=1e-7),
                    "fit.inertia": bool(np.isclose(float(parts[0]["inertia"]), single["inertia"], rtol=1e-9)),
                }
                for name, ok in checks.items():
                    n_checks += 1
                    if not ok:
                        fails.append((name, info))
                        print("FAIL", name, info, flush=True)
                n_cases += 1
        if os.path.exists(rdv):
            os.unlink(rdv)
        print(f"... {n_cases} fits, {n_checks} checks, {len(fails)} failures ({time.time() - t0:.0f} s)", flush=True)
    print(f"sharded fit sweep: seed {seed}, {n_cases} fits, {n_checks} checks, {len(fails)} failures")
    sys.exit(1 if fails else 0)


def main():
    if len(sys.argv) > 3 and sys.argv[3] == "fit":
        main_fit(float(sys.argv[1]), int(sys.argv[2]))
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    sys.path[:0] = [PKG_DIR]
    from pccs_accel import kernels as K
    rs = np.random.RandomState(seed)
    devc = torch.device("cuda:0")
    t0 = time.time()
    n_cases = n_checks = n_reloc = n_empty_rank = 0
    fails = []
    while time.time() - t0 < budget:
        world = int(rs.choice([2, 3, 4, 5, 8]))
        seeds = [int(s) for s in rs.randint(1 << 30, size=CASES_PER_SPAWN)]
        fd, rdv = tempfile.mkstemp(prefix="mtp_rdv_")
        os.close(fd)
        with tempfile.TemporaryDirectory() as out_dir:
            mp.spawn(worker, args=(world, rdv, seeds, out_dir), nprocs=world, join=True)
            for s in seeds:
                X, C0, tol, ce, max_iter = make_case(s)
                bounds = make_bounds(s, X.shape[0], world)
                sl, si, sc, sn = K.lloyd(torch.from_numpy(X).to(devc), torch.from_numpy(C0).to(devc), max_iter=max_iter,
                                         tol_abs=tol, check_every=ce)
                parts = [np.load(os.path.join(out_dir, f"c{s}_r{r}.npz")) for r in range(world)]
                info = dict(seed=s, world=world, n=X.shape[0], d=X.shape[1], k=C0.shape[0], ce=ce, max_iter=max_iter, bounds=bounds)
                got = np.concatenate([p["labels"] for p in parts])
                checks = {
                    "n_iter": all(int(p["n_iter"]) == sn for p in parts),
                    "replicated": all(np.array_equal(p["centers"], parts[0]["centers"]) and
                                      float(p["inertia"]) == float(parts[0]["inertia"]) for p in parts),
                    "labels": np.array_equal(got, sl.cpu().numpy()),
                    "centers": np.allclose(parts[0]["centers"], sc.cpu().numpy(), rtol=1e-6, atol=1e-7),
                    "inertia": bool(np.isclose(float(parts[0]["inertia"]), si, rtol=1e-9)),
                }
                for name, ok in checks.items():
                    n_checks += 1
                    if not ok:
                        fails.append((name, info))
                        print("FAIL", name, info, flush=True)
                n_cases += 1
                n_reloc += int(np.abs(C0).max() > 40.0)
                n_empty_rank += int(any(lo == hi for lo, hi in bounds))
        if os.path.exists(rdv):  # the FileStore removes it when the last rank leaves
            os.unlink(rdv)
        print(f"... {n_cases} cases, {n_checks} checks, {len(fails)} failures ({time.time() - t0:.0f} s)", flush=True)
    print(f"stub sweep: seed {seed}, {n_cases} cases ({n_reloc} with unreachable centres, {n_empty_rank} with a rank "
          f"without rows), {n_checks} checks, {len(fails)} failures")
    sys.exit(1 if fails else 0)


if __name__ == "__main__":
    main()
