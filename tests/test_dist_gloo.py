"""Multi-process (gloo, CPU) tests of the sharded Lloyd driver pccs_accel.dist.ShardedLloyd.

The driver's logic -- shard bounds, the per-iteration all-reduce of the packed double buffer,
replicated M-step, chunked flag checks, the all-gather based empty-cluster relocation, the
final E-step and the inertia reduction -- is exercised with world_size 1, 2 and 3 over gloo.
The HIP kernels are replaced by ``NumpyOps``, a CPU model of the step-level C-ABI built on the
oracle (this is test code; the product's DeviceOps is what runs on a GPU).  Every world size
must reproduce the oracle's single-process ``kmeans_single_lloyd``.
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
import kmeans_oracle as O
from conftest import PKG_DIR, ROOT


class NumpyOps:
    """CPU model of include/mtp_accel.h's step-level entry points (same state machine)."""

    def zeros(self, n, dtype):
        return torch.zeros(n, dtype=dtype)

    def begin(self, C, k, d, staged, c2, red, flags):
        c2.copy_(torch.from_numpy(O.row_sqnorm(C.numpy())))
        red.zero_()
        flags.zero_()

    def assign(self, X, C, staged, c2, k, labels, red, flags):
        if int(flags[0]) != 0:
            return
        d = X.shape[1]
        new = O.assign(X.numpy(), C.numpy(), c2.numpy()) if X.shape[0] else np.zeros(0, np.int32)
        red[k * d + k] += float((new != labels.numpy()).sum())
        labels.copy_(torch.from_numpy(new))

    def accumulate(self, X, k, labels, red, flags):
        if int(flags[0]) != 0 or X.shape[0] == 0:
            return
        d = X.shape[1]
        s, w = O.accumulate_f64(X.numpy(), labels.numpy(), k)
        red[:k * d] += torch.from_numpy(s.ravel())
        red[k * d:k * d + k] += torch.from_numpy(w)

    def mstep(self, red, C_old, C_new, k, d, tol, max_iter, relocation_done, staged, c2, shift, flags, red_backup):
        if int(flags[0]) != 0:
            return
        counts = red[k * d:k * d + k].numpy()
        n_empty = int((counts == 0).sum())
        if n_empty > 0 and not relocation_done:
            red_backup.copy_(red)
            red.zero_()
            flags[0] = 3
            flags[2] = n_empty
            return
        sums = red[:k * d].numpy().reshape(k, d)
        new = np.empty((k, d), dtype=np.float32)
        heavy = int(np.argmax(counts))
        for j in range(k):
            src = j if counts[j] > 0 else heavy
            alpha = np.float32(1.0) if (counts[j] == 0 and src > j) else np.float32(1.0) / np.float32(counts[src])
            new[j] = sums[src].astype(np.float32) * alpha
        sh = np.empty(k, dtype=np.float32)
        old = np.ascontiguousarray(C_old.numpy())
        O.lib().orc_center_shift(O._ptr(old), O._ptr(new), k, d, O._ptr(sh))
        tot = float((sh * sh).astype(np.float64).sum())
        n_changed = float(red[k * d + k])
        C_new.copy_(torch.from_numpy(new))
        c2.copy_(torch.from_numpy(O.row_sqnorm(new)))
        shift.copy_(torch.from_numpy(sh))
        red.zero_()
        it = int(flags[1]) + 1
        flags[1] = it
        flags[2] = 0
        flags[0] = 1 if (n_changed == 0 and tol >= 0) else (2 if tot <= tol else (4 if it >= max_iter else 0))

    def far_points(self, X, C_old, labels, m, idx_offset):
        n, d = X.shape
        val = torch.full((m,), -1.0, dtype=torch.float32)
        idx = torch.full((m,), -1, dtype=torch.int64)
        lab = torch.full((m,), -1, dtype=torch.int32)
        rows = torch.zeros((m, d), dtype=torch.float32)
        if n:
            pp = ((X.numpy() - C_old.numpy()[labels.numpy()]) ** 2).sum(axis=1)  # sklearn's `distances`, NumPy's order
            order = np.lexsort((np.arange(n), -pp))[:m]
            for q, i in enumerate(order):
                val[q], idx[q], lab[q] = float(pp[i]), int(i) + idx_offset, int(labels[i])
                rows[q] = X[i]
        return val, idx, lab, rows

    def apply_relocation(self, red, k, d, m, val, lab, rows):
        if not float(val[0]) > 0:
            return
        counts = red[k * d:k * d + k]
        empties = [j for j in range(k) if float(counts[j]) == 0.0]
        for i in range(min(len(empties), m)):
            new_id, old_id = empties[i], int(lab[i])
            if old_id < 0:
                break
            x = rows[i].to(torch.float64)
            red[old_id * d:(old_id + 1) * d] -= x
            red[new_id * d:(new_id + 1) * d] = x
            counts[new_id] = 1.0
            counts[old_id] -= 1.0

    def final_assign(self, X, C, staged, c2, k, labels):
        if X.shape[0]:
            labels.copy_(torch.from_numpy(O.assign(X.numpy(), C.numpy(), c2.numpy())))

    def snapshot(self, flags):
        return flags.clone()

    def wait(self, snap):
        return snap.numpy()

    def inertia(self, X, C, labels):
        v = O.inertia(X.numpy(), C.numpy(), labels.numpy()) if X.shape[0] else 0.0
        return torch.tensor([v], dtype=torch.float64)


def _case(which):
    if which == "plain":
        X = datagen.make_encodings(14, 1500, 16, 6, 1500, 900)
        Xc = X - X.mean(axis=0)
        C0 = Xc[datagen.make_init_indices(5, 1500, 24)].copy()
    elif which == "relocate":  # an unreachable centre -> empty cluster -> relocation through the all-gather path
        X = datagen.make_encodings(15, 900, 8, 5, 2000, 300)
        Xc = X - X.mean(axis=0)
        C0 = Xc[datagen.make_init_indices(6, 900, 9)].copy()
        C0[2] = 40.0
        C0[7] = -35.0
    else:  # "many_empty": 100 of 160 centres unreachable -> more far points than any rank's share of the old cap of 64
        X = datagen.make_encodings(16, 1300, 8, 20, 2000, 300)
        Xc = X - X.mean(axis=0)
        C0 = Xc[datagen.make_init_indices(9, 1300, 160)].copy()
        for j in range(100):
            C0[(3 * j) % 160] = 30.0 + j
    return Xc, C0, float(O.tolerance(X))


def _oracle(which, Xc, C0, max_iter, tol):
    # several clusters empty at once: the far points are handed out by descending distance (DESIGN.md deviation (iv);
    # sklearn's argpartition order is unspecified), which the oracle is told to do as well
    return O.kmeans_single_lloyd(Xc, C0, max_iter, tol, far_order="descending" if which == "many_empty" else None)


def _worker(rank, world, port, which, check_every, out_dir, fixed_steps=0, max_iter=300):
    sys.path[:0] = [PKG_DIR, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests", "golden")]
    dist.init_process_group("gloo", init_method=f"file://{port}", rank=rank, world_size=world)
    from pccs_accel import dist as pdist
    Xc, C0, tol = _case(which)
    lo, hi = pdist.shard_bounds(Xc.shape[0], rank, world)
    sl = pdist.ShardedLloyd(torch.from_numpy(Xc[lo:hi].copy()), torch.from_numpy(C0.copy()), row_offset=lo,
                            ops=NumpyOps())
    extra = {}
    if fixed_steps:  # the first iterations with the flags read one step late (what bench.py times)
        n_reloc, noop, attempts, state = sl.run_fixed(fixed_steps, tol, max_iter)
        extra = dict(n_reloc=n_reloc, n_noop=len(noop), attempts=attempts, state=state,
                     iters_after_fixed=int(sl.read_flags()[1]))
    labels, inertia, centers, n_iter = sl.run(max_iter=max_iter, tol_abs=tol, check_every=check_every)
    np.savez(os.path.join(out_dir, f"r{rank}.npz"), labels=labels.numpy(), centers=centers.numpy(), inertia=inertia,
             n_iter=n_iter, lo=lo, hi=hi, **extra)
    dist.destroy_process_group()


def _free_port():
    """A rendezvous for one spawn: the path of a file for torch.distributed's FileStore (handed to the workers as
    ``port``).  A TCP port picked here could be taken by someone else before rank 0 binds it (seen on a shared box)."""
    fd, path = tempfile.mkstemp(prefix="mtp_rdv_")
    os.close(fd)
    os.unlink(path)
    return path


@pytest.mark.parametrize("which", ["plain", "relocate", "many_empty"])
@pytest.mark.parametrize("world,check_every", [(1, 1), (2, 1), (2, 5), (3, 2)])
def test_sharded_lloyd_matches_oracle(which, world, check_every, tmp_path):
    mp.spawn(_worker, args=(world, _free_port(), which, check_every, str(tmp_path)), nprocs=world, join=True)
    Xc, C0, tol = _case(which)
    ol, oi, oc, on = _oracle(which, Xc, C0, 300, tol)
    parts = [np.load(tmp_path / f"r{r}.npz") for r in range(world)]
    labels = np.concatenate([p["labels"] for p in parts])
    assert [int(p["lo"]) for p in parts] == sorted(int(p["lo"]) for p in parts) and int(parts[-1]["hi"]) == Xc.shape[0]
    for p in parts:  # replicated state is identical on every rank
        assert int(p["n_iter"]) == on
        assert np.array_equal(p["centers"], parts[0]["centers"])
        assert float(p["inertia"]) == float(parts[0]["inertia"])
    assert np.array_equal(labels, ol)
    assert np.allclose(parts[0]["centers"], oc, rtol=1e-5, atol=1e-6)
    assert np.isclose(float(parts[0]["inertia"]), oi, rtol=1e-6)


@pytest.mark.parametrize("which", ["plain", "relocate", "many_empty"])
@pytest.mark.parametrize("world,fixed_steps", [(1, 3), (2, 3), (2, 1000), (3, 1)])
def test_lagged_flag_checks_match_oracle(which, world, fixed_steps, tmp_path):
    """run_fixed (flags looked at one step late, paused / finished attempts re-enqueued or dropped)
    followed by run() must land on the oracle's result, whether the fixed part ends mid-run (3),
    right behind the relocation pause (1) or runs into convergence (1000)."""
    mp.spawn(_worker, args=(world, _free_port(), which, 1, str(tmp_path), fixed_steps), nprocs=world, join=True)
    Xc, C0, tol = _case(which)
    ol, oi, oc, on = _oracle(which, Xc, C0, 300, tol)
    parts = [np.load(tmp_path / f"r{r}.npz") for r in range(world)]
    labels = np.concatenate([p["labels"] for p in parts])
    for p in parts:
        assert int(p["n_iter"]) == on
        assert np.array_equal(p["centers"], parts[0]["centers"])
        assert int(p["iters_after_fixed"]) == min(fixed_steps, on)
        # every attempt is either an iteration or a recorded no-op
        assert int(p["attempts"]) - int(p["n_noop"]) == min(fixed_steps, on)
        if fixed_steps < on:
            assert int(p["state"]) == 0
        if which == "relocate":
            assert int(p["n_reloc"]) >= 1
        if which == "many_empty":
            assert int(p["n_reloc"]) >= 100
    assert np.array_equal(labels, ol)
    assert np.allclose(parts[0]["centers"], oc, rtol=1e-5, atol=1e-6)
    assert np.isclose(float(parts[0]["inertia"]), oi, rtol=1e-6)


@pytest.mark.parametrize("which", ["plain", "relocate"])
def test_lagged_flag_checks_stop_at_max_iter(which, tmp_path):
    """max_iter reached inside run_fixed: the loop state becomes MAXITER (4), the attempt enqueued behind it is a
    recorded no-op, and run() only adds the final E-step -- the oracle's max_iter=3 result."""
    mp.spawn(_worker, args=(2, _free_port(), which, 1, str(tmp_path), 10, 3), nprocs=2, join=True)
    Xc, C0, tol = _case(which)
    ol, oi, oc, on = O.kmeans_single_lloyd(Xc, C0, 3, tol)
    parts = [np.load(tmp_path / f"r{r}.npz") for r in range(2)]
    assert on == 3
    for p in parts:
        assert int(p["n_iter"]) == 3 and int(p["iters_after_fixed"]) == 3 and int(p["state"]) == 4
        assert int(p["attempts"]) - int(p["n_noop"]) == 3
    assert np.array_equal(np.concatenate([p["labels"] for p in parts]), ol)
    assert np.allclose(parts[0]["centers"], oc, rtol=1e-5, atol=1e-6)
    assert np.isclose(float(parts[0]["inertia"]), oi, rtol=1e-6)


def _eval_worker(rank, world, port, out_dir):
    sys.path[:0] = [PKG_DIR, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests", "golden")]
    dist.init_process_group("gloo", init_method=f"file://{port}", rank=rank, world_size=world)
    from pccs_accel import dist as pdist
    pred, gt = _eval_case()
    lo, hi = pdist.shard_bounds(pred.shape[0], rank, world)

    def cpu_kernel(p, g, mode=0, ade_div=None, best_of=None):  # the oracle in the kernel's place
        o = O.best_of_k(p.numpy(), g.numpy(), mode, ade_div=ade_div, best_of=best_of)
        return dict(min_ade=torch.from_numpy(o["min_ade"]), min_fde=torch.from_numpy(o["min_fde"]),
                    idx=torch.from_numpy(o["idx"]), sums=torch.tensor([o["sum_ade"], o["sum_fde"]], dtype=torch.float64))

    out = {}
    for mode in (0, 1, 2):
        r = pdist.best_of_k_sharded(torch.from_numpy(pred[lo:hi].copy()), torch.from_numpy(gt[lo:hi].copy()),
                                    pred.shape[0], mode=mode, kernel=cpu_kernel)
        out[f"mean{mode}"] = np.array([r["mean_ade"], r["mean_fde"]])
        out[f"idx{mode}"] = r["idx"].numpy()
    np.savez(os.path.join(out_dir, f"e{rank}.npz"), **out)
    dist.destroy_process_group()


def _eval_case():
    rng = np.random.default_rng(5)
    gt = np.cumsum(rng.normal(0, 0.3, (101, 12, 2)), axis=1).astype(np.float32)
    pred = (gt[:, None] + rng.normal(0, 0.2, (101, 20, 12, 2))).astype(np.float32)
    return pred, gt


@pytest.mark.parametrize("world", [1, 2, 3])
def test_best_of_k_sharded_means(world, tmp_path):
    """Agents sharded, one all-reduce of the two sums: every rank reports the global means."""
    mp.spawn(_eval_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    pred, gt = _eval_case()
    parts = [np.load(tmp_path / f"e{r}.npz") for r in range(world)]
    for mode in (0, 1, 2):
        o = O.best_of_k(pred, gt, mode)
        want = np.array([o["mean_ade"], o["mean_fde"]])
        for p in parts:
            assert np.allclose(p[f"mean{mode}"], want, rtol=1e-12)
        assert np.array_equal(np.concatenate([p[f"idx{mode}"] for p in parts]), o["idx"])


class NumpySeedOps:
    """CPU model of the step-level seeding entry points (oracle arithmetic)."""

    def row_sqnorm(self, X):
        return torch.zeros(max(X.shape[0], 1), dtype=torch.float64)

    def count_below(self, closest, thresholds):
        c = closest.numpy()
        out = np.zeros(thresholds.shape, dtype=np.int64)
        for r in range(c.shape[0]):
            out[r] = np.searchsorted(np.cumsum(c[r], dtype=np.float64), thresholds[r].numpy())
        return torch.from_numpy(out)

    def candidate_dists(self, X, yy, cand, closest):
        b, l, _ = cand.shape
        n = X.shape[0]
        rows = np.zeros((b, l, n), dtype=np.float32)
        if n:
            for r in range(b):
                dd = O._euclidean_distances_sq_f32(cand[r].numpy(), X.numpy())
                if closest is not None:
                    np.minimum(closest[r].numpy()[None, :], dd, out=dd)
                rows[r] = dd
        return torch.from_numpy(rows), torch.from_numpy(rows.astype(np.float64).sum(axis=2))


def _seed_case():
    X = datagen.make_encodings(31, 1201, 16, 7, 1500, 900)
    return np.ascontiguousarray(X - X.mean(axis=0))


def _seed_worker(rank, world, port, out_dir, packed=True):
    sys.path[:0] = [PKG_DIR, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests", "golden")]
    dist.init_process_group("gloo", init_method=f"file://{port}", rank=rank, world_size=world)
    from pccs_accel import dist as pdist
    from pccs_accel import seeding
    Xc = _seed_case()
    n, k, b = Xc.shape[0], 12, 3
    rs = np.random.RandomState(17)
    draws = [seeding.draw(rs, n, k) for _ in range(b)]
    lo, hi = pdist.shard_bounds(n, rank, world)
    centers, indices = seeding.kmeans_plusplus_sharded(torch.from_numpy(Xc[lo:hi].copy()), lo, n, k, draws,
                                                       ops=NumpySeedOps(), packed=packed)
    np.savez(os.path.join(out_dir, f"s{rank}.npz"), centers=centers.numpy(), indices=indices.numpy())
    dist.destroy_process_group()


@pytest.mark.parametrize("world,packed", [(1, True), (2, True), (3, True), (2, False), (3, False)])
def test_sharded_seeding_matches_oracle(world, packed, tmp_path):
    """k-means++ over row-sharded points (ranks' cumulative-sum offsets, counted draws, exchanged candidate
    rows, gathered potentials): every rank ends with the seeds sklearn's algorithm picks, restart by
    restart, from the same RandomState stream -- with the two packed all-gathers per centre and with the
    fetch-by-index exchange the packed form falls back to."""
    mp.spawn(_seed_worker, args=(world, _free_port(), str(tmp_path), packed), nprocs=world, join=True)
    Xc = _seed_case()
    rs = np.random.RandomState(17)
    want = [O.kmeans_plusplus(Xc, 12, rs, pot="f64") for _ in range(3)]  # the sharded seeding sums in double
    parts = [np.load(tmp_path / f"s{r}.npz") for r in range(world)]
    for p in parts:
        assert np.array_equal(p["indices"], parts[0]["indices"]) and np.array_equal(p["centers"], parts[0]["centers"])
        for r in range(3):
            assert np.array_equal(p["indices"][r], want[r][1])
            assert np.array_equal(p["centers"][r], want[r][0])


class BlasSeedOps(NumpySeedOps):
    """NumpySeedOps plus the piecewise OpenBLAS-order sums (oracle.c's restatements of sgemv_t / sdot)."""

    def block_sums(self, rows, pot64=None, eps=0.0):
        r = rows.numpy()
        out = np.stack([O.sgemv_t_block_sums(r[i]) for i in range(r.shape[0])])
        if pot64 is not None:  # rows outside the margin are not summed (zeros), like the device entry
            p = pot64.numpy()
            out[~(p * (1.0 - eps) <= p.min(axis=1, keepdims=True) * (1.0 + eps))] = 0.0
        return torch.from_numpy(out)

    def combine(self, parts):
        return torch.from_numpy(O.combine_block_sums(parts.numpy()))

    def sdot_carry(self, x, main_local, left_over, acc, finish):
        a = acc.numpy()  # shares memory: updated in place
        out = np.zeros(a.shape[0], dtype=np.float32)
        for i in range(a.shape[0]):
            out[i] = O.sdot_carry(x[i].numpy() if x.shape[1] else np.zeros(0, np.float32), main_local, left_over, a[i],
                                  finish)
        return torch.from_numpy(out) if finish else None


def _blas_seed_case(n):
    X = datagen.make_encodings(77, n, 16, 5, 1500, 900)
    return np.ascontiguousarray(X - X.mean(axis=0))


def _blas_seed_worker(rank, world, port, out_dir, n):
    sys.path[:0] = [PKG_DIR, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests", "golden")]
    dist.init_process_group("gloo", init_method=f"file://{port}", rank=rank, world_size=world)
    from pccs_accel import dist as pdist
    from pccs_accel import seeding
    Xc = _blas_seed_case(n)
    k, b = 7, 2
    rs = np.random.RandomState(23)
    draws = [seeding.draw(rs, n, k) for _ in range(b)]
    lo, hi = pdist.shard_bounds(n, rank, world, align=seeding.POT_ALIGN)
    centers, indices = seeding.kmeans_plusplus_sharded(torch.from_numpy(Xc[lo:hi].copy()), lo, n, k, draws,
                                                       ops=BlasSeedOps(), force_collectives=True)
    np.savez(os.path.join(out_dir, f"s{rank}.npz"), centers=centers.numpy(), indices=indices.numpy())
    if world > 1:  # a split that is not block-aligned cannot keep the order and says so (on every rank)
        lo, hi = pdist.shard_bounds(n, rank, world)
        with pytest.raises(ValueError, match="multiple of"):
            seeding.kmeans_plusplus_sharded(torch.from_numpy(Xc[lo:hi].copy()), lo, n, k, draws, ops=BlasSeedOps(),
                                            pot_order="openblas")
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n", [(1, 9001), (2, 3 * 4096 + 1003), (3, 3 * 4096 + 1003), (3, 2 * 4096), (2, 4096 + 66)])
def test_sharded_seeding_in_openblas_order(world, n, tmp_path):
    """Row-sharded k-means++ with the float32 potentials added as NumPy + OpenBLAS add them (sgemv_t's 4096-point
    blocks formed per rank and combined in rank order; sdot's 64 accumulators handed from rank to rank for the first
    centre): with block-aligned shards -- an empty last rank and a 2-point sdot tail among the cases -- every rank
    picks the seeds of the oracle's OpenBLAS-order k-means++, which fixture G14 pins to sklearn's own."""
    mp.spawn(_blas_seed_worker, args=(world, _free_port(), str(tmp_path), n), nprocs=world, join=True)
    Xc = _blas_seed_case(n)
    rs = np.random.RandomState(23)
    want = [O.kmeans_plusplus(Xc, 7, rs, pot="blas_model") for _ in range(2)]
    parts = [np.load(tmp_path / f"s{r}.npz") for r in range(world)]
    for p in parts:
        for r in range(2):
            assert np.array_equal(p["indices"][r], want[r][1])
            assert np.array_equal(p["centers"][r], want[r][0])


def _empty_rank_worker(rank, world, port):
    sys.path[:0] = [PKG_DIR]
    dist.init_process_group("gloo", init_method=f"file://{port}", rank=rank, world_size=world)
    from pccs_accel import dist as pdist
    X = torch.zeros((5 if rank == 0 else 0, 16), dtype=torch.float32)
    with pytest.raises(ValueError, match="at least one row"):  # on every rank, before any kernel or further collective
        pdist.kmeans_fit_sharded(X, 5, row_offset=0 if rank == 0 else 5, n_clusters=2, n_init=1, random_state=0,
                                 stats=(None, None, None))
    dist.destroy_process_group()


def test_fit_with_an_empty_rank_is_refused_everywhere():
    mp.spawn(_empty_rank_worker, args=(2, _free_port()), nprocs=2, join=True)


def _fit_worker(rank, world, port, out_dir, numpy_stats=False):
    sys.path[:0] = [PKG_DIR, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests", "golden")]
    dist.init_process_group("gloo", init_method=f"file://{port}", rank=rank, world_size=world)
    from pccs_accel import dist as pdist
    X = _fit_case()
    lo, hi = pdist.shard_bounds(X.shape[0], rank, world)
    stats = (lambda A: A.double().sum(dim=0), lambda A, m: ((A.double() - m) ** 2).sum(dim=0),
             lambda A, v: A.sub_(v))
    seqsum = None
    if numpy_stats:  # the column statistics as float32 chains over all rows, carried from rank to rank
        def seqsum(A, center, init):
            return torch.from_numpy(O.col_seqsum(A.numpy(), None if center is None else center.numpy(), init.numpy()))
    fit = pdist.kmeans_fit_sharded(torch.from_numpy(X[lo:hi].copy()), X.shape[0], row_offset=lo, n_clusters=9, n_init=4,
                                   random_state=5, lloyd_ops=NumpyOps(), seed_ops=NumpySeedOps(), stats=stats, seqsum=seqsum)
    np.savez(os.path.join(out_dir, f"f{rank}.npz"), labels=fit["labels"].numpy(), centers=fit["centers"].numpy(),
             inertia=fit["inertia"], n_iter=fit["n_iter"], tol_abs=fit["tol_abs"])
    dist.destroy_process_group()


def _fit_case():
    return np.ascontiguousarray(datagen.make_encodings(41, 901, 16, 9, 1500, 900))


@pytest.mark.parametrize("world,numpy_stats", [(1, False), (2, False), (3, False), (1, True), (2, True), (3, True)])
def test_sharded_full_fit_matches_oracle(world, numpy_stats, tmp_path):
    """KMeans(n_clusters, n_init=4, random_state=5).fit over sharded points: global mean / tolerance, sharded
    k-means++ for all restarts, ShardedLloyd per restart, sklearn's best-restart rule with the sharded
    same-clustering check -- the oracle's labels, iteration count and inertia."""
    mp.spawn(_fit_worker, args=(world, _free_port(), str(tmp_path), numpy_stats), nprocs=world, join=True)
    X = _fit_case()
    want = O.kmeans_fit(X, 9, n_init=4, random_state=5, pot="f64")
    parts = [np.load(tmp_path / f"f{r}.npz") for r in range(world)]
    if numpy_stats:  # the chains carried over the ranks give NumPy's own tolerance, whatever the number of ranks
        for p in parts:
            assert float(p["tol_abs"]) == float(O.tolerance(X))
    labels = np.concatenate([p["labels"] for p in parts])
    assert np.array_equal(labels, want["labels"])
    for p in parts:
        assert int(p["n_iter"]) == want["n_iter"]
        assert np.allclose(p["centers"], want["centers"], rtol=1e-5, atol=1e-6)
        assert np.isclose(float(p["inertia"]), want["inertia"], rtol=1e-6)


def test_same_clustering_sharded_single_process():
    sys.path.insert(0, PKG_DIR)
    from pccs_accel.dist import is_same_clustering_sharded
    a = torch.tensor([0, 1, 2, 1, 0, 2], dtype=torch.int32)
    assert is_same_clustering_sharded(a, torch.tensor([2, 0, 1, 0, 2, 1], dtype=torch.int32), 3, False)
    assert not is_same_clustering_sharded(a, torch.tensor([2, 0, 1, 1, 2, 1], dtype=torch.int32), 3, False)
    assert is_same_clustering_sharded(a[:0], a[:0], 3, False)


def test_shard_bounds_cover_everything():
    sys.path.insert(0, PKG_DIR)
    from pccs_accel.dist import shard_bounds
    for n in (0, 1, 7, 10_000_000, 73_568):
        for w in (1, 2, 3, 8):
            b = [shard_bounds(n, r, w) for r in range(w)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(w - 1))
            sizes = [e - s for s, e in b]
            assert max(sizes) - min(sizes) <= 1
