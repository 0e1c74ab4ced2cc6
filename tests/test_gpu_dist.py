"""Sharded Lloyd with the real HIP kernels, world_size 2 and 3 on ONE GPU (gloo moves the packed
reduce buffer; on a multi-GPU node the same driver runs over RCCL, one process per GPU).  Each
world size must reproduce the single-process device run (itself checked against the CPU oracle
in test_gpu_parity.py): identical labels and n_iter, centres to 1e-6 -- the double-accumulated
sums make the trajectory independent of how the points are split.
"""
import os
import sys
import tempfile

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import datagen
from conftest import PKG_DIR, ROOT

pytestmark = pytest.mark.gpu


def _case(which):
    if which == "plain":
        X = datagen.make_encodings(21, 30011, 96, 50, 1500, 900)   # ragged: not a multiple of anything
        C0 = X[datagen.make_init_indices(7, 30011, 200)].copy()
    elif which == "relocate":  # unreachable centres -> empty clusters -> relocation through the all-gather path
        X = datagen.make_encodings(22, 9001, 128, 12, 2000, 300)
        C0 = X[datagen.make_init_indices(8, 9001, 40)].copy()
        C0[3] = 40.0
        C0[17] = -35.0
        C0[31] = 25.0
    else:  # "many_empty": 100 of 160 centres unreachable, all relocated in the first iteration (the old cap was 64)
        X = datagen.make_encodings(23, 7003, 96, 30, 2000, 300)
        C0 = X[datagen.make_init_indices(9, 7003, 160)].copy()
        for j in range(100):
            C0[(3 * j) % 160] = 30.0 + j
    return X, C0


def _worker(rank, world, port, which, check_every, out_dir, lagged_steps=0):
    import faulthandler
    faulthandler.dump_traceback_later(60, exit=True)  # a stuck rank names its line and ends the test
    sys.path[:0] = [PKG_DIR, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests", "golden")]
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", init_method=f"file://{port}", rank=rank, world_size=world)
    from pccs_accel import dist as pdist
    X, C0 = _case(which)
    dev = torch.device("cuda:0")
    lo, hi = pdist.shard_bounds(X.shape[0], rank, world)
    fit = pdist.kmeans_fit_sharded(torch.from_numpy(X[lo:hi].copy()).to(dev), X.shape[0], torch.from_numpy(C0).to(dev),
                                   row_offset=lo, check_every=check_every, lagged_steps=lagged_steps)
    np.savez(os.path.join(out_dir, f"r{rank}.npz"), labels=fit["labels"].cpu().numpy(),
             centers=fit["centers"].cpu().numpy(), inertia=fit["inertia"], n_iter=fit["n_iter"], lo=lo, hi=hi)
    dist.destroy_process_group()
    faulthandler.cancel_dump_traceback_later()


def _free_port():
    """A rendezvous for one spawn: the path of a file for torch.distributed's FileStore (handed to the workers as
    ``port``).  A TCP port picked here could be taken by someone else before rank 0 binds it (seen on a shared box)."""
    fd, path = tempfile.mkstemp(prefix="mtp_rdv_")
    os.close(fd)
    os.unlink(path)
    return path


@pytest.mark.parametrize("which", ["plain", "relocate", "many_empty"])
@pytest.mark.parametrize("world,check_every,lagged", [(2, 1, 0), (3, 4, 0), (2, 1, 5), (1, 1, 1000)])
def test_sharded_device_lloyd_matches_single(which, world, check_every, lagged, tmp_path, gpu_device):
    """lagged > 0: the first iterations go through run_fixed (flags read one step late through
    pinned snapshots, what bench.py times); 1000 runs it into convergence."""
    from pccs_accel import clustering
    mp.spawn(_worker, args=(world, _free_port(), which, check_every, str(tmp_path), lagged), nprocs=world, join=True)
    X, C0 = _case(which)
    single = clustering.kmeans_fit(torch.from_numpy(X).to(gpu_device), C0.shape[0], init=torch.from_numpy(C0))
    parts = [np.load(tmp_path / f"r{r}.npz") for r in range(world)]
    labels = np.concatenate([p["labels"] for p in parts])
    for p in parts:
        assert int(p["n_iter"]) == single["n_iter"]
        assert np.array_equal(p["centers"], parts[0]["centers"])
        assert float(p["inertia"]) == float(parts[0]["inertia"])
    assert np.array_equal(labels, single["labels"].cpu().numpy())
    assert np.allclose(parts[0]["centers"], single["centers"].cpu().numpy(), rtol=1e-6, atol=1e-7)
    assert np.isclose(float(parts[0]["inertia"]), single["inertia"], rtol=1e-9)


def _rccl_worker(rank, port, which, out_dir):
    import faulthandler
    faulthandler.dump_traceback_later(90, exit=True)
    sys.path[:0] = [PKG_DIR, os.path.join(ROOT, "tests", "golden")]
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dev = torch.device("cuda:0")
    dist.init_process_group("nccl", init_method=f"file://{port}", rank=0, world_size=1, device_id=dev)
    from pccs_accel import dist as pdist
    X, C0 = _case(which)
    fit = pdist.kmeans_fit_sharded(torch.from_numpy(X).to(dev), X.shape[0], torch.from_numpy(C0).to(dev),
                                   row_offset=0, check_every=2, force_collectives=True, lagged_steps=4)
    np.savez(os.path.join(out_dir, "rccl.npz"), labels=fit["labels"].cpu().numpy(),
             centers=fit["centers"].cpu().numpy(), inertia=fit["inertia"], n_iter=fit["n_iter"])
    dist.destroy_process_group()
    faulthandler.cancel_dump_traceback_later()


@pytest.mark.parametrize("which", ["plain", "relocate", "many_empty"])
def test_rccl_one_rank_collectives(which, tmp_path, gpu_device):
    """The RCCL ("nccl") backend itself: a one-rank communicator on the box's one GPU, every
    collective of the sharded driver issued on the device buffers (fp64 all-reduce of the packed
    sums, the relocation all-gathers, the inertia all-reduce).  Same answer as the plain run."""
    from pccs_accel import clustering
    mp.spawn(_rccl_worker, args=(_free_port(), which, str(tmp_path)), nprocs=1, join=True)
    X, C0 = _case(which)
    single = clustering.kmeans_fit(torch.from_numpy(X).to(gpu_device), C0.shape[0], init=torch.from_numpy(C0))
    p = np.load(tmp_path / "rccl.npz")
    assert int(p["n_iter"]) == single["n_iter"]
    assert np.array_equal(p["labels"], single["labels"].cpu().numpy())
    assert np.allclose(p["centers"], single["centers"].cpu().numpy(), rtol=1e-6, atol=1e-7)
    assert np.isclose(float(p["inertia"]), single["inertia"], rtol=1e-9)


def _native_worker(rank, port, which, out_dir):
    import faulthandler
    faulthandler.dump_traceback_later(90, exit=True)
    sys.path[:0] = [PKG_DIR, os.path.join(ROOT, "tests", "golden")]
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dev = torch.device("cuda:0")
    dist.init_process_group("nccl", init_method=f"file://{port}", rank=0, world_size=1, device_id=dev)
    from pccs_accel import dist as pdist
    X, C0 = _case(which)
    comm = pdist.RcclComm(dev)
    fit = pdist.kmeans_fit_sharded(torch.from_numpy(X).to(dev), X.shape[0], torch.from_numpy(C0).to(dev), row_offset=0,
                                   check_every=2, comm=comm)
    # and once with per-phase timing and without the closing E-step (what bench.py calls)
    Xc = torch.from_numpy(X).to(dev) - torch.from_numpy(X.mean(axis=0, dtype=np.float64).astype(np.float32)).to(dev)
    C0c = torch.from_numpy(C0).to(dev) - torch.from_numpy(X.mean(axis=0, dtype=np.float64).astype(np.float32)).to(dev)
    lab5, _, _, it5, phase = pdist.lloyd_sharded_native(comm, Xc.contiguous(), C0c.contiguous(), 0, X.shape[0], max_iter=5,
                                                        tol_abs=-1.0, check_every=5, want_phase_ms=True, final_estep=False)
    comm.close()
    np.savez(os.path.join(out_dir, "native.npz"), labels=fit["labels"].cpu().numpy(), centers=fit["centers"].cpu().numpy(),
             inertia=fit["inertia"], n_iter=fit["n_iter"], it5=it5, phase=np.asarray(phase))
    dist.destroy_process_group()
    faulthandler.cancel_dump_traceback_later()


@pytest.mark.parametrize("which", ["plain", "relocate", "many_empty"])
def test_in_library_rccl_loop(which, tmp_path, gpu_device):
    """mtp_lloyd_sharded_f32: the whole Lloyd loop inside the library with its own RCCL communicator
    (mtp_comm_create: unique id from rank 0, ncclCommInitRank) -- per iteration an in-stream ncclAllReduce of
    the packed sums, ncclAllGather + merge of the far points for empty clusters, ncclAllReduce of the inertia.
    One rank on the box's one GPU (RCCL refuses two ranks on one device): same answer as the one-GPU fit."""
    from pccs_accel import clustering
    mp.spawn(_native_worker, args=(_free_port(), which, str(tmp_path)), nprocs=1, join=True)
    X, C0 = _case(which)
    single = clustering.kmeans_fit(torch.from_numpy(X).to(gpu_device), C0.shape[0], init=torch.from_numpy(C0))
    p = np.load(tmp_path / "native.npz")
    assert int(p["n_iter"]) == single["n_iter"]
    assert np.array_equal(p["labels"], single["labels"].cpu().numpy())
    assert np.array_equal(p["centers"], single["centers"].cpu().numpy())
    assert float(p["inertia"]) == single["inertia"]
    assert int(p["it5"]) == 5     # tol_abs < 0: exactly max_iter iterations, no convergence test
    assert p["phase"].shape == (4,) and (p["phase"] >= 0).all() and p["phase"][0] > 0


def _pp_case(n=6007):
    X = datagen.make_encodings(51, n, 96, 24, 1500, 900)
    return np.ascontiguousarray(X)


def _pp_worker(rank, world, port, backend, out_dir, n=6007, order="f64"):
    import faulthandler
    faulthandler.dump_traceback_later(120, exit=True)
    sys.path[:0] = [PKG_DIR, os.path.join(ROOT, "tests", "golden")]
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dev = torch.device("cuda:0")
    if backend == "nccl":
        dist.init_process_group("nccl", init_method=f"file://{port}", rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group("gloo", init_method=f"file://{port}", rank=rank, world_size=world)
    from pccs_accel import dist as pdist
    from pccs_accel import seeding
    X = _pp_case(n)
    n, k, b = X.shape[0], 24, 5
    lo, hi = pdist.shard_bounds(n, rank, world, align=seeding.POT_ALIGN if order == "openblas" else 1)
    Xl = torch.from_numpy(X[lo:hi].copy()).to(dev)
    # seeding alone, on data centred with the global mean (what the fit passes it)
    Xc = torch.from_numpy((X - X.mean(axis=0, dtype=np.float64).astype(np.float32))[lo:hi].copy()).to(dev)
    rs = np.random.RandomState(23)
    draws = [seeding.draw(rs, n, k) for _ in range(b)]
    centers, indices = seeding.kmeans_plusplus_sharded(Xc, lo, n, k, draws, force_collectives=True, pot_order=order)
    fit = pdist.kmeans_fit_sharded(Xl, n, row_offset=lo, n_clusters=k, n_init=b, random_state=23, force_collectives=True,
                                   pot_order=order)
    np.savez(os.path.join(out_dir, f"p{rank}.npz"), seed_centers=centers.cpu().numpy(), seed_indices=indices.cpu().numpy(),
             labels=fit["labels"].cpu().numpy(), centers=fit["centers"].cpu().numpy(), inertia=fit["inertia"],
             n_iter=fit["n_iter"])
    dist.destroy_process_group()
    faulthandler.cancel_dump_traceback_later()


@pytest.mark.parametrize("backend,world,n,order", [("gloo", 2, 6007, "f64"), ("gloo", 3, 6007, "f64"),
                                                   ("nccl", 1, 6007, "f64"), ("nccl", 1, 6007, "openblas"),
                                                   ("gloo", 2, 3 * 4096 + 1003, "openblas"),
                                                   ("gloo", 3, 2 * 4096 + 70, "openblas")])
def test_sharded_seeding_and_full_fit_match_single_gpu(backend, world, n, order, tmp_path, gpu_device):
    """k-means++ and the whole KMeans(n_clusters, n_init).fit over row-sharded points with the real kernels
    (mtp_kpp_candidate_dists_f32 + the step-level Lloyd entries) against the one-GPU path: same seeds for every
    restart, same labels / iteration count, centres and inertia to rounding -- with double potentials (any split)
    and with the float32 potentials in the reference stack's order (4096-aligned shards: block sums per rank combined
    in rank order, the sdot accumulators handed on), which is the one-GPU default that fixture G14 pins to sklearn."""
    from pccs_accel import clustering, seeding
    mp.spawn(_pp_worker, args=(world, _free_port(), backend, str(tmp_path), n, order), nprocs=world, join=True)
    X = _pp_case(n)
    n, k, b = X.shape[0], 24, 5
    Xc = torch.from_numpy(X - X.mean(axis=0, dtype=np.float64).astype(np.float32)).to(gpu_device)
    rs = np.random.RandomState(23)
    draws = [seeding.draw(rs, n, k) for _ in range(b)]
    c1, i1 = seeding.kmeans_plusplus_batched(Xc, k, draws, pot_order=order)
    single = clustering.kmeans_fit(torch.from_numpy(X).to(gpu_device), k, n_init=b, random_state=23, pot_order=order)
    parts = [np.load(tmp_path / f"p{r}.npz") for r in range(world)]
    for p in parts:
        assert np.array_equal(p["seed_indices"], i1.cpu().numpy())
        assert np.array_equal(p["seed_centers"], c1.cpu().numpy())
        assert int(p["n_iter"]) == single["n_iter"]
        assert np.allclose(p["centers"], single["centers"].cpu().numpy(), rtol=1e-6, atol=1e-7)
        assert np.isclose(float(p["inertia"]), single["inertia"], rtol=1e-9)
    assert np.array_equal(np.concatenate([p["labels"] for p in parts]), single["labels"].cpu().numpy())
