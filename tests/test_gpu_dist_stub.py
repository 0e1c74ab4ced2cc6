"""The in-library sharded Lloyd loop (mtp_lloyd_sharded_f32, csrc/capi.hip lloyd_run) with world_size 2 and 3.

RCCL refuses two ranks on one device and the test box has one GPU, so the ranks are PROCESSES sharing cuda:0 and
the library binds its six collective entry points from tests/stub_nccl/libstubnccl.so (MTP_RCCL_LIB; a
shared-memory exchange, sums in rank order) instead of librccl.so.  Everything else is the product path: the
library's own communicator object, its per-iteration all-reduce of the packed double sums in stream order, the
parked buffer + zero all-reduces behind a relocation pause, the four all-gathers + k_far_merge with row offsets, a
rank without rows, the inertia all-reduce.  Each run must reproduce the one-GPU mtp_lloyd_f32 run: identical
labels and iteration count, centres identical on every rank and equal to the one-GPU ones to 1e-6 (double sums
added in another order), inertia to 1e-9.
"""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import datagen
from conftest import PKG_DIR, ROOT
from test_gpu_dist import _case, _free_port

pytestmark = pytest.mark.gpu

STUB = os.path.join(ROOT, "tests", "stub_nccl", "libstubnccl.so")


def _bounds(n, world, split):
    if split == "even":
        from pccs_accel import dist as pdist
        return [pdist.shard_bounds(n, r, world) for r in range(world)]
    if split == "empty_last":      # the last rank holds no rows
        cut = [0] + [n * (r + 1) // (world - 1) for r in range(world - 1)] + [n]
        return [(cut[r], cut[r + 1]) for r in range(world)]
    if split == "empty_first":     # rank 0 holds no rows, its all-gather slot is all fillers
        cut = [0, 0] + [n * (r + 1) // (world - 1) for r in range(world - 1)]
        return [(cut[r], cut[r + 1]) for r in range(world)]
    raise ValueError(split)


def _stub_worker(rank, world, port, which, check_every, split, out_dir):
    import faulthandler
    faulthandler.dump_traceback_later(110, exit=True)
    os.environ["MTP_RCCL_LIB"] = STUB           # read by the library at its first collective call
    sys.path[:0] = [PKG_DIR, os.path.join(ROOT, "tests", "golden")]
    torch.cuda.set_device(0)
    dev = torch.device("cuda:0")
    dist.init_process_group("gloo", init_method=f"file://{port}", rank=rank, world_size=world)
    from pccs_accel import dist as pdist
    X, C0 = _case(which)
    mean = X.mean(axis=0, dtype=np.float64).astype(np.float32)
    Xc, C0c = X - mean, C0 - mean
    tol = float(np.mean(np.var(X, axis=0)) * 1e-4)
    lo, hi = _bounds(X.shape[0], world, split)[rank]
    comm = pdist.RcclComm(dev)                  # the 128-byte id travels over gloo; mtp_comm_create on every rank
    labels, inertia, centers, n_iter = pdist.lloyd_sharded_native(
        comm, torch.from_numpy(np.ascontiguousarray(Xc[lo:hi])).to(dev), torch.from_numpy(C0c).to(dev), lo, X.shape[0],
        max_iter=300, tol_abs=tol, check_every=check_every)
    comm.close()
    np.savez(os.path.join(out_dir, f"s{rank}.npz"), labels=labels.cpu().numpy(), centers=centers.cpu().numpy(),
             inertia=inertia, n_iter=n_iter, lo=lo, hi=hi)
    dist.destroy_process_group()
    faulthandler.cancel_dump_traceback_later()


@pytest.mark.parametrize("which,world,check_every,split", [
    ("plain", 2, 1, "even"), ("plain", 3, 4, "even"),
    ("relocate", 2, 4, "even"),          # pause inside a window: parked buffer, zero all-reduces, re-enqueue
    ("relocate", 3, 1, "even"),
    ("many_empty", 2, 1, "even"),        # 100 far points per rank: all-gathers + merge with row offsets
    ("many_empty", 3, 4, "empty_last"),  # a rank with n_local == 0 (k_fill_no_far_points, no E-step launches)
    ("relocate", 3, 2, "empty_first"),
    ("plain", 2, 1000, "even"),          # the whole run enqueued at once
])
def test_library_loop_over_stub_collectives(which, world, check_every, split, tmp_path, gpu_device):
    from pccs_accel import kernels as K
    assert os.path.exists(STUB), "tests/stub_nccl/libstubnccl.so is not built (python -c 'import __graft_entry__ as g; g.build()')"
    mp.spawn(_stub_worker, args=(world, _free_port(), which, check_every, split, str(tmp_path)), nprocs=world, join=True)
    X, C0 = _case(which)
    mean = X.mean(axis=0, dtype=np.float64).astype(np.float32)
    tol = float(np.mean(np.var(X, axis=0)) * 1e-4)
    sl, si, sc, sn = K.lloyd(torch.from_numpy(X - mean).to(gpu_device), torch.from_numpy(C0 - mean).to(gpu_device), tol_abs=tol)
    parts = [np.load(tmp_path / f"s{r}.npz") for r in range(world)]
    assert [(int(p["lo"]), int(p["hi"])) for p in parts] == _bounds(X.shape[0], world, split)
    for p in parts:
        assert int(p["n_iter"]) == sn
        assert np.array_equal(p["centers"], parts[0]["centers"])          # replicated bit for bit, no broadcast
        assert float(p["inertia"]) == float(parts[0]["inertia"])
    assert np.array_equal(np.concatenate([p["labels"] for p in parts]), sl.cpu().numpy())
    assert np.allclose(parts[0]["centers"], sc.cpu().numpy(), rtol=1e-6, atol=1e-7)
    assert np.isclose(float(parts[0]["inertia"]), si, rtol=1e-9)
    assert len(np.unique(sl.cpu().numpy())) == C0.shape[0]               # every relocated cluster ended up populated


def test_far_merge_against_lexsort(gpu_device):
    """k_far_merge alone (mtp_far_merge_f32): three ranks' far-point lists with equal distances across ranks, `-1`
    fillers from a rank with fewer points than m and from a rank without rows, against NumPy's lexsort top-m
    (descending distance, ties to the lowest global row index)."""
    from pccs_accel._lib import check, lib, ptr, stream_ptr
    rng = np.random.RandomState(5)
    for m, d, world in [(7, 96, 3), (1, 128, 2), (40, 8, 4), (5, 96, 3)]:
        total = world * m
        val = rng.choice(np.array([0.5, 1.25, 1.25, 2.0, 3.5, 3.5, 9.0], dtype=np.float32), size=total)  # many ties
        idx = rng.permutation(10 * total)[:total].astype(np.int64)
        lab = rng.randint(0, 50, size=total).astype(np.int32)
        rows = rng.randn(total, d).astype(np.float32)
        # rank 1 offers only two points, the last rank none; the (5, 96, 3) case has fewer points than m in total
        fill = np.zeros(total, dtype=bool)
        fill[m + min(2, m):2 * m] = True
        fill[(world - 1) * m:] = True
        if (m, world) == (5, 3):
            fill[2:m] = True
        val[fill], idx[fill], lab[fill] = -1.0, -1, -1
        rows[fill] = 0
        for r in range(world):  # each rank's own list is in descending order, as mtp_far_points_f32 writes it
            seg = slice(r * m, (r + 1) * m)
            o = np.lexsort((np.where(idx[seg] < 0, np.iinfo(np.int64).max, idx[seg]), -val[seg]))
            val[seg], idx[seg], lab[seg], rows[seg] = val[seg][o], idx[seg][o], lab[seg][o], rows[seg][o]
        g = [torch.from_numpy(a).to(gpu_device) for a in (val, idx, lab, rows)]
        out_val = torch.empty(m, dtype=torch.float32, device=gpu_device)
        out_idx = torch.empty(m, dtype=torch.int64, device=gpu_device)
        out_lab = torch.empty(m, dtype=torch.int32, device=gpu_device)
        out_rows = torch.empty((m, d), dtype=torch.float32, device=gpu_device)
        failed = torch.full((1,), 7, dtype=torch.int32, device=gpu_device)
        check(lib().mtp_far_merge_f32(ptr(g[0]), ptr(g[1]), ptr(g[2]), ptr(g[3]), total, m, d, ptr(out_val), ptr(out_idx),
                                      ptr(out_lab), ptr(out_rows), ptr(failed), stream_ptr()), "mtp_far_merge_f32")
        real = np.nonzero(idx >= 0)[0]
        order = real[np.lexsort((idx[real], -val[real]))][:m]
        exp_val = np.full(m, -1.0, dtype=np.float32)
        exp_idx = np.full(m, -1, dtype=np.int64)
        exp_lab = np.full(m, -1, dtype=np.int32)
        exp_rows = np.zeros((m, d), dtype=np.float32)
        exp_val[:len(order)], exp_idx[:len(order)], exp_lab[:len(order)], exp_rows[:len(order)] = \
            val[order], idx[order], lab[order], rows[order]
        assert np.array_equal(out_val.cpu().numpy(), exp_val)
        assert np.array_equal(out_idx.cpu().numpy(), exp_idx)
        assert np.array_equal(out_lab.cpu().numpy(), exp_lab)
        assert np.array_equal(out_rows.cpu().numpy(), exp_rows)
        assert int(failed.item()) == 0
        # a rank whose search failed marks its list with -2: every rank sees it
        g[0][m] = -2.0
        check(lib().mtp_far_merge_f32(ptr(g[0]), ptr(g[1]), ptr(g[2]), ptr(g[3]), total, m, d, ptr(out_val), ptr(out_idx),
                                      ptr(out_lab), ptr(out_rows), ptr(failed), stream_ptr()), "mtp_far_merge_f32")
        assert int(failed.item()) == 1
