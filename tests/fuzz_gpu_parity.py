#!/usr/bin/env python3
"""Randomised parity sweep: the HIP path (through the C-ABI) against the CPU oracle on seeded random shapes,
for a bounded time.  Development aid, run on the MI355X box:

    python tests/fuzz_gpu_parity.py [seconds] [seed]
    python tests/fuzz_gpu_parity.py replay <case function> <sub-seed> [<sub-seed> ...]

Not collected by pytest (the fixed-shape tests in tests/test_gpu_*.py are the suite); lives under tests/ because it
imports the oracle, which is test infrastructure only.  Same bars as the suite: integer outputs bit-exact, sums to
1e-12, centres / inertia / ADE / FDE to 1e-5 relative.  Prints one line per failing case and a summary; exit code 1
if anything failed."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("modular-trajectory-prediction-toolkit_amd", "oracle", os.path.join("tests", "golden")):
    sys.path.insert(0, os.path.join(ROOT, p))

import datagen  # noqa: E402
import glue_oracle as G  # noqa: E402
import kmeans_oracle as O  # noqa: E402
from pccs_accel import kernels as K  # noqa: E402
from pccs_accel import seeding  # noqa: E402
from pccs_accel import clustering, scene  # noqa: E402

DEV = torch.device("cuda:0")
FAIL = []
COUNT = {}


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def check(cond, case, **info):
    COUNT[case] = COUNT.get(case, 0) + 1
    if not cond:
        FAIL.append((case, info))
        print("FAIL", case, info, flush=True)


def pick_d(rs):
    return int(rs.choice([64, 96, 128, 64, 96, 128, int(rs.randint(1, 41)), 4 * int(rs.randint(1, 60))]))


def enc(rs, n, d, k_true, spread, noise):
    """datagen's mixture (integers / 4096), or -- every other time -- the same scaled and jittered into floats
    with full mantissas, so that no product or sum is exact by construction."""
    X = datagen.make_encodings(int(rs.randint(1 << 30)), n, d, k_true, spread, noise)
    if rs.rand() < 0.5:
        X = X * np.float32(rs.uniform(0.3, 3.0)) + (rs.standard_normal(X.shape) * 1e-3).astype(np.float32)
    return np.ascontiguousarray(X.astype(np.float32))


def case_assign(rs):
    n, d, k = int(rs.randint(1, 6000)), pick_d(rs), int(rs.randint(1, 320))
    X = enc(rs, n, d, max(1, k // 2), 1500, 700)
    C = X[rs.randint(0, n, size=k)].copy()
    if rs.rand() < 0.3:  # exact duplicates among the centres: the lowest index must win
        C[rs.randint(0, k)] = C[rs.randint(0, k)]
    prev_h = rs.randint(-1, k, size=n).astype(np.int32)
    prev = dev(prev_h)
    nch = torch.zeros(1, dtype=torch.float64, device=DEV)
    got = K.assign(dev(X), dev(C), labels=prev, n_changed=nch).cpu().numpy()
    want = O.assign(X, C, backend="chain")
    check(np.array_equal(got, want), "assign", n=n, d=d, k=k, mism=int((got != want).sum()))
    check(int(nch.item()) == int((want != prev_h).sum()), "assign.n_changed", n=n, d=d, k=k)


def case_sums(rs):
    n, d, k = int(rs.randint(1, 8000)), pick_d(rs), int(rs.randint(1, 1500))
    X = enc(rs, n, d, 5, 1500, 700)
    labels = rs.randint(0, k, size=n).astype(np.int32)
    if rs.rand() < 0.3:
        labels[:] = labels[0]  # everything in one cluster
    sums, counts = K.centroid_sums(dev(X), dev(labels), k)
    ws, wc = O.accumulate_f64(X, labels, k)
    check(np.array_equal(counts.cpu().numpy(), wc), "sums.counts", n=n, d=d, k=k)
    check(np.allclose(sums.cpu().numpy(), ws, rtol=1e-12, atol=1e-12), "sums", n=n, d=d, k=k)


def case_lloyd(rs):
    n, d, k = int(rs.randint(40, 2500)), int(rs.choice([64, 96, 128, 12, 20])), int(rs.randint(1, 40))
    k = min(k, n)
    X = enc(rs, n, d, max(1, k), 1500, 700)
    X = X - X.mean(0)
    C0 = X[rs.choice(n, size=k, replace=False)].copy()
    tol = float(O.tolerance(X)) if rs.rand() < 0.7 else 0.0
    ce = int(rs.choice([1, 3, 8]))
    labels, inertia, centers, n_iter = K.lloyd(dev(X), dev(C0), max_iter=60, tol_abs=tol, check_every=ce)
    ol, oi, oc, on = O.kmeans_single_lloyd(X, C0, 60, tol, sums="f64")
    info = dict(n=n, d=d, k=k, tol=tol, check_every=ce)
    check(n_iter == on, "lloyd.n_iter", got=n_iter, want=on, **info)
    check(np.array_equal(labels.cpu().numpy(), ol), "lloyd.labels", mism=int((labels.cpu().numpy() != ol).sum()), **info)
    check(np.allclose(centers.cpu().numpy(), oc, rtol=1e-5, atol=1e-6), "lloyd.centers", **info)
    check(np.isclose(inertia, oi, rtol=1e-5), "lloyd.inertia", got=inertia, want=float(oi), **info)
    # the batched path, the same run three times over with one different start
    if k >= 2 and d in (64, 96, 128) and rs.rand() < 0.5:  # the batched entry takes the MFMA feature sizes only
        C1 = X[rs.choice(n, size=k, replace=False)].copy()
        bl, bi, bc, bn = K.lloyd_batched(dev(X), dev(np.stack([C0, C1, C0])), max_iter=60, tol_abs=tol, check_every=ce)
        check(bn[0] == on and bn[2] == on, "lloyd_batched.n_iter", got=bn, want=on, **info)
        check(np.array_equal(bl[0].cpu().numpy(), ol) and np.array_equal(bl[2].cpu().numpy(), ol), "lloyd_batched.labels", **info)
        l1, i1, c1, n1 = K.lloyd(dev(X), dev(C1), max_iter=60, tol_abs=tol, check_every=ce)
        check(bn[1] == n1 and torch.equal(bl[1], l1) and torch.equal(bc[1], c1), "lloyd_batched.second", **info)


def case_best_of_k(rs):
    n, k = int(rs.randint(1, 3000)), int(rs.randint(1, 130))
    t = int(rs.choice([12, 12, 12, 8, 5, 1, 20]))
    dt = np.float32 if rs.rand() < 0.7 else np.float64
    pred, gt = datagen.make_trajectories(int(rs.randint(1 << 30)), n, k, t, dtype=dt)
    if k > 1 and rs.rand() < 0.5:  # exact ties: a duplicated candidate
        a = rs.randint(0, n, size=max(1, n // 4))
        pred[a, rs.randint(0, k)] = pred[a, rs.randint(0, k)]
    mode = int(rs.randint(0, 3))
    best_of = None if rs.rand() < 0.6 else int(rs.randint(1, k + 1))
    g = K.best_of_k(dev(pred), dev(gt), mode=mode, best_of=best_of)
    o = O.best_of_k(pred, gt, mode=mode, best_of=best_of)
    info = dict(n=n, k=k, t=t, dt=dt.__name__, mode=mode, best_of=best_of)
    check(np.array_equal(g["idx"].cpu().numpy(), o["idx"]), "best_of_k.idx", **info)
    check(np.array_equal(g["idx2"].cpu().numpy(), o["idx2"]), "best_of_k.idx2", **info)
    check(np.allclose(g["min_ade"].cpu().numpy(), o["min_ade"], rtol=1e-5, atol=0), "best_of_k.min_ade", **info)
    check(np.allclose(g["min_fde"].cpu().numpy(), o["min_fde"], rtol=1e-5, atol=0), "best_of_k.min_fde", **info)
    s = g["sums"].cpu().numpy()
    check(np.isclose(s[0], o["sum_ade"], rtol=1e-9) and np.isclose(s[1], o["sum_fde"], rtol=1e-9), "best_of_k.sums", **info)


def case_decode(rs):
    n, k, t = int(rs.randint(1, 600)), int(rs.randint(1, 40)), int(rs.choice([12, 12, 8, 7, 1]))
    disp = datagen.make_displacements(int(rs.randint(1 << 30)), n, k, t)
    th = rs.rand(n).astype(np.float32) * 6.28 - 3.14
    rot = np.stack([np.stack([np.cos(th), -np.sin(th)], -1), np.stack([np.sin(th), np.cos(th)], -1)], 1).astype(np.float32)
    last = rs.randn(n, 2).astype(np.float32)
    use_rot = rs.rand() < 0.6
    info = dict(n=n, k=k, t=t, rot=use_rot)
    if use_rot:
        got = K.decode_trajectories(dev(disp), dev(rot), dev(last)).cpu().numpy()
        want = G.decode_predictions(disp, rot, last)
        check(np.array_equal(got, want), "decode.rot", maxdiff=float(np.abs(got - want).max()), **info)
    else:
        got = K.decode_trajectories(dev(disp)).cpu().numpy()
        want = G.transform_trajectory(disp)
        check(np.array_equal(got, want), "decode.cumsum", **info)
    # fused displacement scoring == decode then score
    gt = datagen.make_trajectories(int(rs.randint(1 << 30)), n, 1, t)[1]
    mode = int(rs.randint(0, 3))
    a = K.best_of_k_disp(dev(disp), dev(gt), mode=mode)
    b = K.best_of_k(K.decode_trajectories(dev(disp)), dev(gt), mode=mode)
    check(torch.equal(a["idx"], b["idx"]) and torch.equal(a["min_ade"], b["min_ade"]) and torch.equal(a["min_fde"], b["min_fde"]),
          "best_of_k_disp", mode=mode, **info)


def case_nms(rs):
    n, m, t = int(rs.randint(1, 200)), int(rs.randint(2, 64)), 12
    dt = np.float64 if rs.rand() < 0.6 else np.float32
    pred, _, sc32 = datagen.make_candidate_sets(int(rs.randint(1 << 30)), n, m, t, dtype=dt)
    scores = sc32.astype(np.float64) if rs.rand() < 0.5 else sc32
    max_samples = int(rs.randint(1, m + 1))
    best_of = int(rs.randint(1, max_samples + 1))
    thresh = float(rs.choice([0.1, 0.3, 1.0]))
    bin_size = int(rs.choice([2, 1, 3, 5]))
    order, n_sel = K.nms_rerank(dev(pred), dev(scores), max_samples, best_of, bin_size, thresh)
    worder, wsel, wout = G.nms_rerank(pred, scores, max_samples, best_of, bin_size, thresh)
    info = dict(n=n, m=m, dt=dt.__name__, sdt=scores.dtype.name, max_samples=max_samples, best_of=best_of, thresh=thresh)
    check(np.array_equal(order.cpu().numpy(), worder), "nms_rerank.order", **info)
    check(np.array_equal(n_sel.cpu().numpy(), wsel), "nms_rerank.n_selected", **info)
    check(np.array_equal(K.gather_candidates(dev(pred), order).cpu().numpy(), wout), "gather_candidates", **info)


def case_relocation(rs):
    """Centres far outside the data start empty: sklearn's relocation (far points in descending distance here)."""
    n, d, k = int(rs.randint(60, 2500)), int(rs.choice([64, 96, 128, 12])), int(rs.randint(3, 40))
    X = enc(rs, n, d, max(1, k // 2), 1500, 700)
    X = X - X.mean(0)
    C0 = X[rs.choice(n, size=k, replace=False)].copy()
    n_far = int(rs.randint(1, k))
    C0[rs.choice(k, size=n_far, replace=False)] += np.float32(50.0)
    tol = float(O.tolerance(X))
    ce = int(rs.choice([1, 4]))
    labels, inertia, centers, n_iter = K.lloyd(dev(X), dev(C0), max_iter=40, tol_abs=tol, check_every=ce)
    ol, oi, oc, on = O.kmeans_single_lloyd(X, C0, 40, tol, sums="f64", far_order="descending")
    info = dict(n=n, d=d, k=k, n_far=n_far, check_every=ce)
    check(n_iter == on, "relocation.n_iter", got=n_iter, want=on, **info)
    check(np.array_equal(labels.cpu().numpy(), ol), "relocation.labels", mism=int((labels.cpu().numpy() != ol).sum()), **info)
    check(np.allclose(centers.cpu().numpy(), oc, rtol=1e-5, atol=1e-6), "relocation.centers", **info)
    check(np.isclose(inertia, oi, rtol=1e-5), "relocation.inertia", **info)


def case_batched_relocation(rs):
    """Restarts of one batch that pause for a relocation at different iterations, beside restarts that never do."""
    n, d, k = int(rs.randint(200, 4000)), int(rs.choice([64, 96, 128])), int(rs.randint(3, 60))
    b = int(rs.choice([2, 4, 7, 12]))
    X = enc(rs, n, d, max(1, k // 2), 1500, 700)
    X = X - X.mean(0)
    tol = float(O.tolerance(X))
    C0s = np.stack([X[rs.choice(n, size=k, replace=False)] for _ in range(b)])
    for r in range(b):
        if rs.rand() < 0.6:
            C0s[r, rs.choice(k, size=int(rs.randint(1, k)), replace=False)] += np.float32(50.0)
    ce = int(rs.choice([1, 4]))
    bl, bi, bc, bn = K.lloyd_batched(dev(X), dev(C0s), max_iter=30, tol_abs=tol, check_every=ce)
    bl, bi, bc = bl.cpu().numpy(), bi.cpu().numpy(), bc.cpu().numpy()
    ok_iter = ok_lab = ok_c = ok_in = True
    for r in range(b):
        ol, oi, oc, on = O.kmeans_single_lloyd(X, C0s[r], 30, tol, sums="f64", far_order="descending")
        ok_iter &= (bn[r] == on)
        ok_lab &= bool(np.array_equal(bl[r], ol))
        ok_c &= bool(np.allclose(bc[r], oc, rtol=1e-5, atol=1e-6))
        ok_in &= bool(np.isclose(bi[r], oi, rtol=1e-6))
    info = dict(n=n, d=d, k=k, restarts=b, check_every=ce)
    check(ok_iter, "batched_relocation.n_iter", got=bn, **info)
    check(ok_lab, "batched_relocation.labels", **info)
    check(ok_c, "batched_relocation.centers", **info)
    check(ok_in, "batched_relocation.inertia", **info)


def case_seeding(rs):
    n, d = int(rs.randint(50, 9000)), int(rs.choice([64, 96, 128, 12, 20, 32]))
    k = int(rs.randint(1, min(n, 48) + 1))
    X = enc(rs, n, d, max(2, k), 1500, 700)
    X = X - X.mean(0)
    seed = int(rs.randint(1 << 30))
    _, idx = seeding.kmeans_plusplus(dev(X), k, np.random.RandomState(seed))
    _, widx = O.kmeans_plusplus(X, k, np.random.RandomState(seed))
    info = dict(n=n, d=d, k=k, seed=seed)
    got = idx.cpu().numpy()
    check(np.array_equal(got, widx), "seeding.indices", first_diff=int(np.argmax(got != widx)) if (got != widx).any() else -1, **info)
    # three restarts in shared launches == one at a time, from the same stream of draws
    r1, r2 = np.random.RandomState(seed), np.random.RandomState(seed)
    nb = int(rs.choice([2, 3, 5, 20]))
    draws = [seeding.draw(r1, n, k) for _ in range(nb)]
    _, bidx = seeding.kmeans_plusplus_batched(dev(X), k, draws)
    singles = [seeding.kmeans_plusplus(dev(X), k, r2)[1] for _ in range(nb)]
    check(all(torch.equal(bidx[i], singles[i]) for i in range(nb)), "seeding.batched", restarts=nb, **info)


def case_one_thread_mode(rs):
    """sums="sklearn_1t": float32 row-order chains, float32 shift sum and inertia chain -- sklearn on one thread."""
    n, d, k = int(rs.randint(40, 2000)), int(rs.choice([64, 96, 128, 20, 200])), int(rs.randint(1, 30))
    X = enc(rs, n, d, max(1, k), 1500, 700)
    X = X - X.mean(0)
    C0 = X[rs.choice(n, size=k, replace=False)].copy()
    if k >= 3 and rs.rand() < 0.4:
        C0[rs.choice(k, size=int(rs.randint(1, k - 1)), replace=False)] += np.float32(50.0)  # empty at first: relocation
    tol = float(O.tolerance(X))
    labels, inertia, centers, n_iter = K.lloyd(dev(X), dev(C0), max_iter=40, tol_abs=tol, check_every=1, sums="sklearn_1t")
    ol, oi, oc, on = O.kmeans_single_lloyd(X, C0, 40, tol, sums="f32", far_order="descending", inertia_order="f32_chain")
    info = dict(n=n, d=d, k=k)
    check(n_iter == on, "one_thread.n_iter", got=n_iter, want=on, **info)
    check(np.array_equal(labels.cpu().numpy(), ol), "one_thread.labels", **info)
    check(np.array_equal(centers.cpu().numpy(), oc), "one_thread.centers_bits", **info)
    check(np.float32(inertia) == np.float32(oi), "one_thread.inertia_bits", got=float(inertia), want=float(oi), **info)


def case_targets(rs):
    n, k = int(rs.randint(1, 5000)), int(rs.randint(1, 450))
    labels = rs.randint(0, k, size=n).astype(np.int32)
    ll, oh = K.label_targets(dev(labels), k)
    wl, wo = O.classifier_targets(labels, k)
    check(np.array_equal(ll.cpu().numpy(), np.asarray(wl)) and ll.dtype == torch.int64, "targets.long", n=n, k=k)
    check(np.array_equal(oh.cpu().numpy(), np.asarray(wo)), "targets.one_hot", n=n, k=k)


def case_large(rs):
    """Sizes that fill the chip: several rounds of E-step workgroups plus the split-launch tail, cluster ranges in the
    sums, and a 20-restart batched fit through the restart-grouped E-step (every restart against the oracle)."""
    d = int(rs.choice([64, 96, 128]))
    n, k = int(rs.randint(100_000, 400_000)), int(rs.randint(100, 1100))
    X = enc(rs, n, d, 64, 1500, 700)
    C = X[rs.randint(0, n, size=k)].copy()
    got = K.assign(dev(X), dev(C)).cpu().numpy()
    want = O.assign(X, C, backend="chain")
    check(np.array_equal(got, want), "large.assign", n=n, d=d, k=k, mism=int((got != want).sum()))
    sums, counts = K.centroid_sums(dev(X), dev(want), k)
    ws, wc = O.accumulate_f64(X, want, k)
    check(np.array_equal(counts.cpu().numpy(), wc) and np.allclose(sums.cpu().numpy(), ws, rtol=1e-12, atol=1e-12),
          "large.sums", n=n, d=d, k=k)
    nb, kb, b = int(rs.randint(5_000, 30_000)), int(rs.randint(20, 230)), int(rs.choice([5, 12, 20]))
    Xb = enc(rs, nb, d, kb, 1500, 700)
    Xb = Xb - Xb.mean(0)
    tol = float(O.tolerance(Xb))
    C0s = np.stack([Xb[rs.choice(nb, size=kb, replace=False)] for _ in range(b)])
    bl, bi, bc, bn = K.lloyd_batched(dev(Xb), dev(C0s), max_iter=25, tol_abs=tol, check_every=4)
    bl, bi = bl.cpu().numpy(), bi.cpu().numpy()
    ok_iter = ok_lab = ok_in = True
    for r in range(b):
        ol, oi, oc, on = O.kmeans_single_lloyd(Xb, C0s[r], 25, tol, sums="f64", far_order="descending")
        ok_iter &= (bn[r] == on)
        ok_lab &= bool(np.array_equal(bl[r], ol))
        ok_in &= bool(np.isclose(bi[r], oi, rtol=1e-6))
    info = dict(n=nb, d=d, k=kb, restarts=b)
    check(ok_iter, "large.batched.n_iter", **info)
    check(ok_lab, "large.batched.labels", **info)
    check(ok_in, "large.batched.inertia", **info)


def case_fit(rs):
    """Whole seeded fits: column statistics in NumPy's order, k-means++ x n_init, Lloyd, sklearn's restart rule."""
    n, d = int(rs.randint(60, 3000)), int(rs.choice([64, 96, 128, 20, 23, 7]))
    k = int(rs.randint(1, min(n, 24) + 1))
    n_init = int(rs.randint(1, 5))
    X = enc(rs, n, d, max(1, k), 1500, 700)
    seed = int(rs.randint(1 << 30))
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        fit = clustering.kmeans_fit(dev(X), k, n_init=n_init, random_state=seed, sums="sklearn_1t")
        # sklearn on one thread: float32 sums in row order, float32 inertia chain (its order decides the restart rule
        # when two restarts end within a rounding error of each other)
        ofit = O.kmeans_fit(X, k, n_init=n_init, random_state=seed, sums="f32", inertia_order="f32_chain", far_order="descending")
    info = dict(n=n, d=d, k=k, n_init=n_init, seed=seed)
    check(fit["n_iter"] == ofit["n_iter"], "fit.n_iter", got=fit["n_iter"], want=ofit["n_iter"], **info)
    check(np.array_equal(fit["labels"].cpu().numpy(), ofit["labels"]), "fit.labels", **info)
    check(np.allclose(fit["centers"].cpu().numpy(), ofit["centers"], rtol=1e-6, atol=1e-6), "fit.centers", **info)
    check(np.isclose(fit["inertia"], ofit["inertia"], rtol=1e-5), "fit.inertia", **info)
    # column statistics: NumPy's own bits
    tol_abs, mean32 = clustering.tolerance_and_mean(dev(X), 1e-4)
    check(np.array_equal(mean32.cpu().numpy(), X.mean(axis=0)), "fit.mean_bits", **info)
    check(tol_abs == float(np.mean(np.var(X, axis=0)) * 1e-4), "fit.tol_bits", **info)


def case_scene(rs):
    """Scene neighbour lists and soft labels against the restated process_scene / gen_prob."""
    sizes = [int(rs.randint(1, 120)) for _ in range(int(rs.randint(1, 5)))]
    n = sum(sizes)
    trj = np.concatenate([datagen.make_scene_trajectories(int(rs.randint(1 << 30)), sz) for sz in sizes])
    marks = np.concatenate([[0], np.cumsum(sizes)])
    obs_len = int(rs.choice([8, 2, 20]))
    rad = float(rs.choice([0.5, 1.0, 2.0, 6.0]))
    offsets, ids = scene.scene_neighbors(trj, marks, obs_len, rad)
    want = G.process_scene(trj, marks, obs_len, rad)
    off_h, ids_h = offsets.cpu().numpy(), ids.cpu().numpy()
    ok = all(ids_h[off_h[i]:off_h[i + 1]].tolist() == want[i] for i in range(n))
    info = dict(n=n, scenes=len(sizes), obs_len=obs_len, rad=rad)
    check(ok, "scene.neighbours", **info)
    k = int(rs.randint(1, 40))
    labels = rs.randint(0, k, size=n).astype(np.int32)
    dense = scene.soft_labels(offsets, ids, labels, k).cpu().numpy()
    rc, rp = G.gen_prob(want, labels)
    check(np.array_equal(dense, G.scene_soft_labels(rc, rp, k)), "scene.soft_labels", k=k, **info)


def case_reranked(rs):
    n, m, t = int(rs.randint(1, 300)), int(rs.randint(2, 80)), 12
    dt = np.float64 if rs.rand() < 0.5 else np.float32
    pred, gt, _ = datagen.make_candidate_sets(int(rs.randint(1 << 30)), n, m, t, dtype=dt)
    keep = int(rs.randint(1, m + 1))
    order = np.stack([rs.permutation(m)[:keep] for _ in range(n)]).astype(np.int32)
    best_of = int(rs.randint(1, keep + 1))
    mode = int(rs.randint(0, 3))
    g = K.best_of_k_reranked(dev(pred), dev(gt), dev(order), mode=mode, best_of=best_of)
    re = np.take_along_axis(pred, order[:, :, None, None].astype(np.int64), axis=1)
    o = O.best_of_k(np.ascontiguousarray(re), gt, mode=mode, best_of=best_of)
    info = dict(n=n, m=m, keep=keep, best_of=best_of, mode=mode, dt=dt.__name__)
    check(np.array_equal(g["idx"].cpu().numpy(), o["idx"]), "reranked.idx", **info)
    check(np.allclose(g["min_ade"].cpu().numpy(), o["min_ade"], rtol=1e-5, atol=0), "reranked.min_ade", **info)
    check(np.allclose(g["min_fde"].cpu().numpy(), o["min_fde"], rtol=1e-5, atol=0), "reranked.min_fde", **info)


def case_seeding_big_k(rs):
    """More local trials (L = 2 + ln k: 8 from k = 404, 9 from k = 1097: beyond the 8-row candidate tile)."""
    k = int(rs.choice([410, 700, 1100, 1250]))
    n, d = int(rs.randint(k + 10, 4000)), int(rs.choice([64, 96, 128, 20]))
    X = enc(rs, n, d, 40, 1500, 700)
    X = X - X.mean(0)
    seed = int(rs.randint(1 << 30))
    _, idx = seeding.kmeans_plusplus(dev(X), k, np.random.RandomState(seed))
    _, widx = O.kmeans_plusplus(X, k, np.random.RandomState(seed))
    got = idx.cpu().numpy()
    check(np.array_equal(got, widx), "seeding_big_k.indices", n=n, d=d, k=k, seed=seed,
          first_diff=int(np.argmax(got != widx)) if (got != widx).any() else -1)


def case_fit_default(rs):
    """The default mode (double sums) restart by restart against the oracle's same variant: seeds, n_iter, labels."""
    n, d = int(rs.randint(60, 2500)), int(rs.choice([64, 96, 128, 20, 23, 7]))
    k = int(rs.randint(1, min(n, 24) + 1))
    n_init = int(rs.randint(1, 4))
    X = enc(rs, n, d, max(1, k), 1500, 700)
    seed = int(rs.randint(1 << 30))
    import warnings
    restarts = []
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        clustering.kmeans_fit(dev(X), k, n_init=n_init, random_state=seed, restarts_out=restarts)
    Xc = X - X.mean(axis=0)
    tol = float(O.tolerance(X))
    r2 = np.random.RandomState(seed)
    info = dict(n=n, d=d, k=k, n_init=n_init, seed=seed)
    for b in range(n_init):
        c0, idx = O.kmeans_plusplus(Xc, k, r2)
        ol, oi, oc, on = O.kmeans_single_lloyd(Xc, c0, 300, tol, sums="f64", far_order="descending")
        check(np.array_equal(restarts[b]["indices"].cpu().numpy(), idx), "fit_default.seeds", restart=b, **info)
        check(restarts[b]["n_iter"] == on, "fit_default.n_iter", restart=b, got=restarts[b]["n_iter"], want=on, **info)
        check(np.array_equal(restarts[b]["labels"].cpu().numpy(), ol), "fit_default.labels", restart=b, **info)
        check(np.isclose(restarts[b]["inertia"], oi, rtol=1e-6), "fit_default.inertia", restart=b, **info)


def case_dropins(rs):
    """The reference-named entry points on small inputs: metrics.py, get_modality, cluster_data with np.random.seed."""
    from pccs_accel.utils.metrics import displacement_error, final_displacement_error
    from pccs_accel.utils.utility import cluster_data, get_modality
    b, t = int(rs.randint(1, 40)), int(rs.choice([12, 8, 1]))
    pred, gt = datagen.make_trajectories(int(rs.randint(1 << 30)), 1, b, t)
    p, g = torch.from_numpy(pred[0]), torch.from_numpy(gt)           # [b, t, 2], [1, t, 2] like train_steps.py:219-220
    ade = displacement_error(p.to(DEV), g.to(DEV)).cpu().numpy()
    fde = final_displacement_error(p[:, -1].to(DEV), g[:, -1].to(DEV)).cpu().numpy()
    check(np.allclose(ade, O.displacement_error(pred[0], gt), rtol=1e-6, atol=0), "dropin.displacement_error", b=b, t=t)
    check(np.allclose(fde, O.final_displacement_error(pred[0][:, -1], gt[:, -1]), rtol=1e-6, atol=0), "dropin.final_displacement_error", b=b, t=t)
    n, d, k = int(rs.randint(30, 400)), int(rs.choice([96, 64, 20])), int(rs.randint(1, 7))
    X = enc(rs, n, d, max(1, k), 1500, 700)
    labels = rs.randint(0, k, size=n).astype(np.int32)
    labels[:k] = np.arange(k)                                         # no empty cluster (KeyError otherwise, tested elsewhere)
    mod = get_modality(torch.from_numpy(X), labels, k).numpy()
    check(np.allclose(mod, O.get_modality(X, labels, k), rtol=1e-6, atol=1e-7), "dropin.get_modality", n=n, d=d, k=k)
    seed = int(rs.randint(1 << 30))
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        np.random.seed(seed)
        try:
            got = cluster_data(X.copy(), k)
        except KeyError as e:
            got = ("KeyError", e.args)
        np.random.seed(seed)
        try:
            want = O.cluster_data(X.copy(), k)
        except KeyError as e:
            want = ("KeyError", e.args)
    if isinstance(want, tuple) and want and isinstance(want[0], str) or isinstance(got, tuple) and got and isinstance(got[0], str):
        check(got == want, "dropin.cluster_data.keyerror", n=n, d=d, k=k, seed=seed)
    else:
        # the reference's sums are float32 (and 8-threaded), the default mode's double: the same labels, or -- where two
        # restarts end within a rounding error of each other -- another restart's labelling of equal quality
        def quality(lab):
            c = O.get_modality(X, lab, k)
            return float(((X.astype(np.float64) - c[lab]) ** 2).sum())
        same = np.array_equal(got[0], want[0])
        check(same or np.isclose(quality(got[0]), quality(want[0]), rtol=1e-4), "dropin.cluster_data.labels",
              mism=int((got[0] != want[0]).sum()), n=n, d=d, k=k, seed=seed)
        check(np.array_equal(got[1], np.bincount(got[0], minlength=k)), "dropin.cluster_data.sizes", n=n, k=k)
        check(got[0].dtype == np.int32 and got[1].dtype == np.int64 and int(got[1].sum()) == n, "dropin.cluster_data.dtypes", n=n, k=k)


def case_primitives(rs):
    """The small entry points of the fit's pre- and post-processing, each against NumPy / the oracle."""
    n, d = int(rs.randint(1, 5000)), pick_d(rs)
    X = enc(rs, n, d, 5, 1500, 700)
    info = dict(n=n, d=d)
    check(np.array_equal(K.row_sqnorm(dev(X)).cpu().numpy(), np.einsum("ij,ij->i", X, X)), "prim.row_sqnorm_bits", **info)
    X64 = X.astype(np.float64)
    cs = K.col_sums(dev(X))
    check(np.allclose(cs.cpu().numpy(), X64.sum(axis=0), rtol=1e-12, atol=1e-12), "prim.col_sums", **info)
    mean64 = X64.mean(axis=0)
    sq = K.col_sqdev(dev(X), dev(mean64))
    check(np.allclose(sq.cpu().numpy(), ((X64 - mean64) ** 2).sum(axis=0), rtol=1e-11, atol=1e-12), "prim.col_sqdev", **info)
    vec = mean64.astype(np.float32)
    check(np.array_equal(K.sub_rowvec_(dev(X), dev(vec)).cpu().numpy(), X - vec), "prim.sub_rowvec_bits", **info)
    k = int(rs.randint(1, 300))
    labels = rs.randint(0, k, size=n).astype(np.int32)
    sums, counts = K.centroid_sums(dev(X), dev(labels), k)
    means, n_empty = K.cluster_means(sums, counts)
    hs, hc = sums.cpu().numpy(), counts.cpu().numpy()
    full = hc > 0
    check(int(n_empty.item()) == int((~full).sum()), "prim.cluster_means.n_empty", k=k, **info)
    check(np.array_equal(means.cpu().numpy()[full], (hs[full] / hc[full, None]).astype(np.float32)), "prim.cluster_means_bits", k=k, **info)
    # is_same_clustering: a relabelling, and the same with one point moved
    perm = rs.permutation(k).astype(np.int32)
    l2 = perm[labels]
    check(clustering.is_same_clustering(dev(labels), dev(l2), k) is True, "prim.same_clustering", k=k, **info)
    if n >= 2 and k >= 2:
        l3 = l2.copy()
        j = int(rs.randint(n))
        l3[j] = (l3[j] + 1 + rs.randint(k - 1)) % k
        want = O.is_same_clustering(labels, l3, k)
        check(clustering.is_same_clustering(dev(labels), dev(l3), k) == want, "prim.same_clustering.moved", k=k, want=want, **info)
    # per-candidate ADE / FDE
    na, ka, t = int(rs.randint(1, 800)), int(rs.randint(1, 40)), int(rs.choice([12, 8, 1, 20]))
    pred, gt = datagen.make_trajectories(int(rs.randint(1 << 30)), na, ka, t, dtype=np.float32)
    div = float(rs.choice([1.0, float(t)]))
    best_of = None if rs.rand() < 0.5 else int(rs.randint(1, ka + 1))
    ga, gf = K.ade_fde(dev(pred), dev(gt), ade_div=div, best_of=best_of)
    oa, of = O.ade_fde(pred, gt, ade_div=div, best_of=best_of)
    check(np.allclose(ga.cpu().numpy(), oa, rtol=1e-5, atol=0), "prim.ade", n=na, k=ka, t=t, best_of=best_of)
    check(np.allclose(gf.cpu().numpy(), of, rtol=1e-5, atol=0), "prim.fde", n=na, k=ka, t=t, best_of=best_of)


def case_noalloc(rs):
    """mtp_assign_noalloc_f32 against mtp_assign_f32, two calls in a row on the shared built-in buffer."""
    from pccs_accel._lib import lib, ptr, stream_ptr
    for _ in range(2):
        d = pick_d(rs)
        k = int(rs.randint(1, 1500))
        while int(lib().mtp_assign_workspace_bytes(k, d)) > (1 << 20):
            k = max(1, k // 2)
        n = int(rs.randint(1, 6000))
        X = enc(rs, n, d, max(1, k // 2), 1500, 700)
        C = X[rs.randint(0, n, size=k)].copy()
        prev = rs.randint(0, k, size=n).astype(np.int32)
        Xd, Cd = dev(X), dev(C)
        want, got = dev(prev), dev(prev)
        wch = torch.zeros(1, dtype=torch.float64, device=DEV)
        gch = torch.zeros(1, dtype=torch.float64, device=DEV)
        K.assign(Xd, Cd, labels=want, n_changed=wch)
        rc = lib().mtp_assign_noalloc_f32(ptr(Xd), n, d, ptr(Cd), k, ptr(got), ptr(gch), stream_ptr())
        torch.cuda.synchronize()
        check(rc == 0 and torch.equal(got, want) and float(gch.item()) == float(wch.item()), "noalloc.assign", n=n, d=d, k=k, rc=rc)


CASES = [case_assign, case_sums, case_lloyd, case_best_of_k, case_decode, case_nms, case_relocation, case_seeding,
         case_one_thread_mode, case_targets, case_fit, case_scene, case_reranked, case_fit_default, case_dropins,
         case_batched_relocation, case_primitives, case_noalloc]
RARE = [case_large, case_seeding_big_k]  # every tenth round


def replay(case_name, sub_seeds):
    """python tests/fuzz_gpu_parity.py replay <case function> <sub-seed> [...]: one case again from the seed of its own
    RandomState (a failing case's seed is found by walking the main stream on the CPU: the draws do not depend on
    the GPU's results)."""
    fn = globals()[case_name]
    for sub in sub_seeds:
        fn(np.random.RandomState(int(sub)))
    print(f"replay {case_name}: {sum(COUNT.values())} checks, {len(FAIL)} failures")
    sys.exit(1 if FAIL else 0)


def main():
    if len(sys.argv) > 3 and sys.argv[1] == "replay":
        replay(sys.argv[2], sys.argv[3:])
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    rs = np.random.RandomState(seed)
    t0 = time.time()
    rounds = 0
    last_print = t0
    while time.time() - t0 < budget:
        for c in CASES + (RARE if rounds % 10 == 0 else []):
            sub = np.random.RandomState(rs.randint(1 << 31))
            try:
                c(sub)
            except Exception as exc:  # noqa: BLE001  (an unexpected refusal is a finding too)
                FAIL.append((c.__name__, repr(exc)))
                print("EXC", c.__name__, repr(exc), flush=True)
        rounds += 1
        if time.time() - last_print > 20:
            last_print = time.time()
            print(f"... {rounds} rounds, {sum(COUNT.values())} checks, {len(FAIL)} failures", flush=True)
    print(f"fuzz: seed {seed}, {rounds} rounds, {sum(COUNT.values())} checks over {len(COUNT)} kinds, {len(FAIL)} failures")
    for kname in sorted(COUNT):
        print(f"  {kname}: {COUNT[kname]}")
    sys.exit(1 if FAIL else 0)


if __name__ == "__main__":
    main()
