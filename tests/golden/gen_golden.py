"""Generate tests/golden/*.npz from the reference itself.

Run ONLY in the build container (needs /root/reference and scikit-learn 1.7.2):

    python tests/golden/gen_golden.py

It imports the reference's modules from /root/reference (never copies them) and
scikit-learn -- the third-party package that holds the reference's k-means arithmetic
(PCCSNet/utils/utility.py:121) -- runs them on the integer-built inputs of datagen.py and
stores the expected outputs.  The fixtures contain data only.
"""
import argparse
import importlib
import os
import sys
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import datagen  # noqa: E402

REF = "/root/reference"


def _import_ref(subdir):
    """Import <REF>/<subdir>/utils/{utility,metrics}.py under a clean ``utils`` namespace."""
    for m in [m for m in sys.modules if m == "utils" or m.startswith("utils.")]:
        del sys.modules[m]
    path = os.path.join(REF, subdir)
    sys.path.insert(0, path)
    try:
        utility = importlib.import_module("utils.utility")
        metrics = importlib.import_module("utils.metrics")
    finally:
        sys.path.remove(path)
    return utility, metrics


def save(name, **kw):
    p = os.path.join(HERE, name + ".npz")
    np.savez_compressed(p, **kw)
    print("wrote", p, os.path.getsize(p), "bytes")


G1_CASES = [
    # name, seed, n, d, k_true, K, spread, noise
    ("g1_n2048_d96_k32", 11, 2048, 96, 32, 32, 2600, 160),
    ("g1_n6000_d96_k200", 12, 6000, 96, 150, 200, 2600, 160),
    ("g1_n4096_d128_k64", 13, 4096, 128, 48, 64, 2600, 160),
    # overlapping modes, K >> k_true: ~20 Lloyd iterations with many close calls
    ("g1_hard_n5000_d96_k64", 14, 5000, 96, 8, 64, 1500, 900),
]


def gen_g1(utility):
    from sklearn.cluster import KMeans
    from sklearn.cluster._k_means_lloyd import lloyd_iter_chunked_dense
    from sklearn.cluster._kmeans import _tolerance

    for name, seed, n, d, k_true, K, spread, noise in G1_CASES:
        X = datagen.make_encodings(seed, n, d, k_true, spread, noise)
        idx0 = datagen.make_init_indices(seed + 1000, n, K)
        C0 = X[idx0].copy()
        km = KMeans(n_clusters=K, init=C0, n_init=1).fit(X)
        # one raw Lloyd iteration on the centred data, one thread (deterministic float sums)
        x_mean = X.mean(axis=0)
        Xc = X - x_mean
        C0c = C0 - x_mean
        c_new = np.zeros_like(C0c)
        w = np.zeros(K, dtype=np.float32)
        lab1 = np.full(n, -1, dtype=np.int32)
        shift = np.zeros(K, dtype=np.float32)
        lloyd_iter_chunked_dense(Xc, np.ones(n, dtype=np.float32), C0c, c_new, w, lab1, shift, 1)
        # G3: the reference's get_modality on the final labels
        mod = utility.get_modality(torch.from_numpy(X), km.labels_, K).numpy()
        save(name, seed=seed, n=n, d=d, k_true=k_true, K=K, spread=spread, noise=noise, idx0=idx0,
             labels=km.labels_.astype(np.int32), centers=km.cluster_centers_, inertia=np.float64(km.inertia_),
             n_iter=km.n_iter_, tol_abs=np.float64(_tolerance(X, 1e-4)), x_mean=x_mean,
             iter1_labels=lab1, iter1_centers=c_new, iter1_weights=w, iter1_shift=shift,
             modality=mod)


def gen_g2(utility):
    X = datagen.make_encodings(21, 3000, 96, 20)
    np.random.seed(0)
    labels, sizes = utility.cluster_data(X.copy(), 20)
    save("g2_seeded_cluster_data", seed=21, n=3000, d=96, k_true=20, K=20, np_seed=0,
         labels=labels, sizes=sizes)
    # k-means++ alone, on centred data as KMeans.fit calls it
    from sklearn.cluster._kmeans import _kmeans_plusplus
    from sklearn.utils.extmath import row_norms
    Xc = X - X.mean(axis=0)
    _, ind = _kmeans_plusplus(Xc, 20, row_norms(Xc, squared=True), np.ones(3000, dtype=np.float32),
                              np.random.RandomState(5))
    save("g9_kmeanspp", seed=21, n=3000, d=96, k_true=20, K=20, rs_seed=5, indices=ind)


def gen_g4_g5():
    from sklearn.cluster import KMeans
    # sklearn/cluster/tests/test_k_means.py:85-111 (relocated clusters), float64 like the KAT
    X = np.array([[0, 0], [0.5, 0], [0.5, 1], [1, 1]], dtype=np.float64)
    init = np.array([[0.5, 0.5], [3, 3]], dtype=np.float64)
    km = KMeans(n_clusters=2, n_init=1, init=init).fit(X)
    X32 = X.astype(np.float32)
    km32 = KMeans(n_clusters=2, n_init=1, init=init.astype(np.float32)).fit(X32)
    save("g4_relocated_clusters", X=X, init=init, labels=km.labels_, centers=km.cluster_centers_,
         inertia=km.inertia_, n_iter=km.n_iter_, labels32=km32.labels_, centers32=km32.cluster_centers_,
         inertia32=km32.inertia_, n_iter32=km32.n_iter_)
    # sklearn/cluster/tests/test_k_means.py:62-80 (hand-computed results); the KAT's sample
    # weights [3,1,1,3] are restated as repeated rows (the reference never passes weights)
    Xw = np.array([[0, 0]] * 3 + [[0.5, 0], [0.5, 1]] + [[1, 1]] * 3, dtype=np.float32)
    init5 = np.array([[0, 0], [1, 1]], dtype=np.float32)
    km5 = KMeans(n_clusters=2, n_init=1, init=init5).fit(Xw)
    assert list(km5.labels_) == [0, 0, 0, 0, 1, 1, 1, 1] and km5.n_iter_ == 2
    assert np.allclose(km5.cluster_centers_, [[0.125, 0], [0.875, 1]]) and np.isclose(km5.inertia_, 0.375)
    save("g5_kat_results", X=Xw, init=init5, labels=km5.labels_, centers=km5.cluster_centers_,
         inertia=km5.inertia_, n_iter=km5.n_iter_)


def gen_g6(metrics):
    n, k, t = 64, 20, 12
    pred, gt = datagen.make_trajectories(31, n, k, t)
    # exact tie: agent 5's candidates 3 and 7 are identical and the best -> first index must win
    pred[5, 3] = gt[5] + np.float32(1.0 / 4096)
    pred[5, 7] = pred[5, 3]
    # exact FDE tie with different ADE: agent 9, candidates 2 and 11 share the last step
    pred[9, 2, -1] = gt[9, -1]
    pred[9, 11, -1] = gt[9, -1]
    tp, tg = torch.from_numpy(pred), torch.from_numpy(gt)
    ade = np.empty((n, k), dtype=np.float32)
    fde = np.empty((n, k), dtype=np.float32)
    res = {}
    for mode in (0, 1, 2):
        val = np.zeros(2).astype(float)
        mins = np.empty((n, 2), dtype=np.float32)
        idx = np.empty(n, dtype=np.int64)
        for i in range(n):  # PCCSNet/train_steps.py:214-232 with batch_size 1
            out, g = tp[i], tg[i:i + 1]
            A = metrics.displacement_error(out, g) / t
            F = metrics.final_displacement_error(out[:, -1], g[:, -1])
            ade[i], fde[i] = A.numpy(), F.numpy()
            if mode == 0:
                j = torch.argmin(A)
                a, f = A[j].item(), F[j].item()
            elif mode == 1:
                j = torch.argmin(F)
                a, f = A[j].item(), F[j].item()
            else:
                j = torch.argmin(A)
                a, f = torch.min(A).item(), torch.min(F).item()
            val += [a, f]
            mins[i] = (a, f)
            idx[i] = int(j)
        val /= n
        res[f"mode{mode}_mins"] = mins
        res[f"mode{mode}_idx"] = idx
        res[f"mode{mode}_val"] = val
    ade_sum = metrics.displacement_error(tp[0], tg[0:1], mode="sum").item()
    fde_sum = metrics.final_displacement_error(tp[0][:, -1], tg[0:1][:, -1], mode="sum").item()
    save("g6_ade_fde", seed=31, n=n, k=k, t=t, ade=ade, fde=fde, ade_sum0=ade_sum, fde_sum0=fde_sum, **res)


def gen_g7():
    # StimulusVerification/verify_predictions.py:238-248 is a top-level script (argparse +
    # file loads) and cannot be imported; its "Before Verification" block is evaluated here
    # with the same NumPy calls on [N,200,12,2] float64 predictions, best_of=20.
    n, ktot, t, best_of = 32, 200, 12, 20
    pred, fut = datagen.make_trajectories(41, n, ktot, t, dtype=np.float64)
    out = {}
    for fde_prio in (False, True):
        diff = np.linalg.norm(pred[:, :best_of] - fut[:, None], axis=-1)
        tot_a = tot_f = 0
        idx = np.empty(n, dtype=np.int64)
        for i in range(n):
            j = np.argmin(diff[i, :, -1]) if fde_prio else np.argmin(np.sum(diff[i], axis=-1))
            tot_f += diff[i, j, -1]
            tot_a += np.sum(diff[i, j])
            idx[i] = j
        tag = "fdeprio" if fde_prio else "adeprio"
        out[tag + "_ade"] = tot_a / n / t
        out[tag + "_fde"] = tot_f / n
        out[tag + "_idx"] = idx
    save("g7_verify_best_of_20", seed=41, n=n, ktot=ktot, t=t, best_of=best_of, **out)


def gen_g8():
    # the XOE copy imports itself as ``PCCSNet_XOE.utils...`` from MomentaryPrediction/
    sys.path.insert(0, os.path.join(REF, "MomentaryPrediction"))
    try:
        xoe_utility = importlib.import_module("PCCSNet_XOE.utils.utility")
    finally:
        sys.path.pop(0)
    X = datagen.make_encodings(51, 2500, 96, 16)
    args = argparse.Namespace(pm=0.02, obs_feat_size=48)
    np.random.seed(3)
    labels, sizes = xoe_utility.cluster_data(X.copy(), n_cluster=16, save_dir=None, args=args)
    save("g8_xoe_pm", seed=51, n=2500, d=96, k_true=16, K=16, np_seed=3, pm=0.02, obs_feat_size=48,
         labels=labels, sizes=sizes)


def _import_data_process():
    """PCCSNet/utils/data_process.py under a clean ``utils`` namespace."""
    for m in [m for m in sys.modules if m == "utils" or m.startswith("utils.")]:
        del sys.modules[m]
    path = os.path.join(REF, "PCCSNet")
    sys.path.insert(0, path)
    try:
        return importlib.import_module("utils.data_process")
    finally:
        sys.path.remove(path)


def gen_g10():
    """f3: transform_trajectory (imported) and gen_prediction's rotate-back (train_steps.py:253-255,
    evaluated with the same torch calls per agent; train_steps.py itself needs a GPU to import-run)."""
    dp = _import_data_process()
    n, k, t = 48, 20, 12
    disp = datagen.make_displacements(61, n, k, t)
    rs = np.random.RandomState(62)
    theta = rs.randint(-3141, 3142, size=n) / 1000.0
    rot = np.stack([np.stack([np.cos(theta), -np.sin(theta)], -1), np.stack([np.sin(theta), np.cos(theta)], -1)], 1)
    rot = torch.tensor(rot).float()
    last_obs = torch.from_numpy((rs.randint(-40000, 40001, size=(n, 2)) / 4096.0).astype(np.float32))
    cum4 = dp.transform_trajectory(torch.from_numpy(disp.copy())).numpy()
    cum3 = dp.transform_trajectory(torch.from_numpy(disp[:, 0].copy())).numpy()
    full = []
    for i in range(n):
        net_output = dp.transform_trajectory(torch.from_numpy(disp[i].copy()))
        mat = rot[i:i + 1]
        output = torch.bmm(mat.repeat(k, 1, 1), net_output.transpose(1, 2)).transpose(1, 2)
        output += last_obs[i:i + 1].unsqueeze(0)
        full.append(output.cpu().numpy())
    full = np.stack(full)
    assert full.shape == (n, k, t, 2) and full.dtype == np.float32
    save("g10_decode", seed=61, n=n, k=k, t=t, rot=rot.numpy(), last_obs=last_obs.numpy(), cum4=cum4, cum3=cum3,
         decoded=full)


def gen_g11():
    """f4: verify_predictions.py:251-290 (top-level script, not importable): its re-ranking loop and
    the "After Verification" evaluation evaluated with the same torch / NumPy calls, on float64 and
    float32 prediction sets; the composed permutation is tracked beside the data."""
    n, m_tot, t, best_of, bin_size, nms_thresh = 24, 200, 12, 20, 2, 0.3
    out = {}
    for tag, dtype, max_samples in (("f64", np.float64, 200), ("f32", np.float32, 150)):
        pred, fut, scores = datagen.make_candidate_sets(71, n, m_tot, t, dtype=dtype)
        work = pred.copy()
        order = np.empty((n, max_samples), dtype=np.int32)
        n_sel = np.empty(n, dtype=np.int32)
        for i in range(n):
            vs = scores[i][:max_samples] // bin_size
            vs = torch.tensor(vs).float() - torch.arange(0, len(vs)).float() / (2 * len(vs))
            srt = torch.argsort(vs, descending=True).tolist()
            assert len(set(vs.tolist())) == len(vs)  # no ties: the order is well defined
            work[i, :max_samples] = work[i, :max_samples][srt]
            selected, ignored = [0], []
            for idx in range(1, max_samples):
                if len(selected) >= best_of:
                    ignored.append(idx)
                else:
                    dd = np.linalg.norm(work[i, selected] - work[i, idx:idx + 1], axis=-1)[..., -1]
                    (ignored if np.min(dd) < nms_thresh else selected).append(idx)
            new_order = selected + ignored
            work[i, :max_samples] = work[i, :max_samples][new_order]
            order[i] = np.asarray(srt)[new_order]
            n_sel[i] = len(selected)
        assert all(np.array_equal(work[i, :max_samples], pred[i][order[i]]) for i in range(n))
        res = {}
        for fde_prio in (False, True):
            diff = np.linalg.norm(work[:, :best_of] - fut[:, None], axis=-1)
            tot_a = tot_f = 0
            for i in range(n):
                j = np.argmin(diff[i, :, -1]) if fde_prio else np.argmin(np.sum(diff[i], axis=-1))
                tot_f += diff[i, j, -1]
                tot_a += np.sum(diff[i, j])
            p = "fdeprio" if fde_prio else "adeprio"
            res[f"{tag}_{p}_ade"] = tot_a / n / t
            res[f"{tag}_{p}_fde"] = tot_f / n
        out.update(res)
        out[f"{tag}_order"] = order
        out[f"{tag}_n_sel"] = n_sel
        out[f"{tag}_max_samples"] = max_samples
    save("g11_nms_rerank", seed=71, n=n, m_tot=m_tot, t=t, best_of=best_of, bin_size=bin_size, nms_thresh=nms_thresh,
         **out)


def gen_g12():
    """f2: process_scene and gen_prob imported from PCCSNet/utils/data_process.py; the dense soft
    labels of get_scene_data (:212-218) from their outputs (get_scene_data itself does
    np.array() on ragged lists, which NumPy >= 1.24 refuses)."""
    import tempfile
    dp = _import_data_process()
    sizes = [230, 1, 170, 199]
    trj = np.concatenate([datagen.make_scene_trajectories(81 + i, s) for i, s in enumerate(sizes)])
    split_marks = [0] + list(np.cumsum(sizes))
    n, K = len(trj), 24
    labels = np.random.RandomState(85).randint(0, K, size=n).astype(np.int32)
    args = argparse.Namespace(obs_len=8, pred_len=12, data_from="ethucy", rad=1.0, n_cluster=K)
    train_set = argparse.Namespace(trj_data=torch.from_numpy(trj).float(), split_marks=split_marks)
    with tempfile.TemporaryDirectory() as tmp:
        os.makedirs(os.path.join(tmp, "ethucy", "toy"))
        scene = dp.process_scene("toy", train_set, args, save_dir=tmp)
        res_class, res_prob = dp.gen_prob(scene, labels, tmp)
    dense = np.zeros((n, K)).astype(float)
    for i in range(n):
        dense[i][res_class[i]] = res_prob[i]
    nb_off = np.zeros(n + 1, dtype=np.int64)
    nb_off[1:] = np.cumsum([len(scene[i]) for i in range(n)])
    nb_ids = np.asarray([j for i in range(n) for j in scene[i]], dtype=np.int64)
    cl_off = np.zeros(n + 1, dtype=np.int64)
    cl_off[1:] = np.cumsum([len(c) for c in res_class])
    assert nb_off[-1] > 1500 and (np.diff(nb_off) == 0).any()
    save("g12_scene_soft_labels", seeds=np.asarray([81 + i for i in range(len(sizes))]), sizes=np.asarray(sizes),
         split_marks=np.asarray(split_marks), obs_len=8, rad=1.0, K=K, labels=labels, nb_off=nb_off, nb_ids=nb_ids,
         cl_off=cl_off, classes=np.concatenate(res_class).astype(np.int64), probs=np.concatenate(res_prob),
         dense=dense)


def gen_g13():
    # PCCSNet/train_steps.py:357 and :364-366 are inline statements of train() (which needs a GPU): the label
    # hand-off is evaluated here with the reference's own torch expressions on a small label vector.
    import torch
    rng = np.random.default_rng(61)
    n, k = 53, 12
    cluster_result = rng.integers(0, k, size=n).astype(np.int32)
    cluster_result[:3] = [0, k - 1, 0]
    cluster_results = torch.from_numpy(cluster_result).long()
    gt_prob = torch.zeros(n, k)
    gt_prob[list(range(n)), cluster_result] = 1
    save("g13_classifier_targets", n=n, k=k, cluster_result=cluster_result, cluster_results=cluster_results.numpy(),
         gt_prob=gt_prob.numpy())


G14_CASES = [
    # name, data seed, n, d, k_true, K, spread, noise, np.random.seed
    ("g14_fit_eth_n73568_k200", 141, 73568, 96, 120, 200, 1800, 420, 7),
    ("g14_fit_sdd_n17970_k400", 142, 17970, 96, 150, 400, 1800, 420, 8),
    # the other four leave-one-out splits of ETH/UCY (N_train hotel 76 646, univ 30 128, zara1 74 084, zara2 66 976)
    ("g14_fit_hotel_n76646_k200", 143, 76646, 96, 120, 200, 1800, 420, 9),
    ("g14_fit_univ_n30128_k200", 144, 30128, 96, 120, 200, 1800, 420, 10),
    ("g14_fit_zara1_n74084_k200", 145, 74084, 96, 120, 200, 1800, 420, 11),
    ("g14_fit_zara2_n66976_k200", 146, 66976, 96, 120, 200, 1800, 420, 12),
]


def gen_g14(utility, only=None):
    """The reference's real sizes (PCCSNet/README.md: eth split N=73 568 / K=200, SDD N=17 970 / K=400):
    ``np.random.seed(s); utility.cluster_data(X, K)`` -- KMeans(K, n_init=20).fit -- with sklearn's two per-restart
    functions wrapped (outputs recorded, nothing altered; the random stream is consumed exactly as without
    the wrappers): per restart the k-means++ indices, n_iter, inertia and a digest of the labels, plus the
    winning restart, the final labels and the sizes -- and, from two repetitions of every restart's Lloyd run,
    how far sklearn's own outcome moves from run to run (rerun_n_iter, rerun_label_diff)."""
    import zlib
    import sklearn.cluster._kmeans as skm

    for name, seed, n, d, k_true, K, spread, noise, np_seed in G14_CASES:
        if only and name not in only:
            continue
        X = datagen.make_encodings(seed, n, d, k_true, spread, noise)
        rec = dict(idx=[], n_iter=[], inertia=[], crc=[], sizes=[], labels=[])
        kpp0, lloyd0 = skm._kmeans_plusplus, skm._kmeans_single_lloyd

        def kpp(*a, **kw):
            centers, indices = kpp0(*a, **kw)
            rec["idx"].append(np.asarray(indices, dtype=np.int64).copy())
            return centers, indices

        def lloyd(*a, **kw):
            labels, inertia, centers, n_iter = lloyd0(*a, **kw)
            rec["n_iter"].append(n_iter)
            rec["inertia"].append(float(inertia))
            rec["crc"].append(zlib.crc32(np.ascontiguousarray(labels, dtype=np.int32).tobytes()))
            rec["sizes"].append(np.bincount(labels, minlength=K).astype(np.int32))
            rec["labels"].append(np.array(labels, dtype=np.int32))
            return labels, inertia, centers, n_iter

        skm._kmeans_plusplus, skm._kmeans_single_lloyd = kpp, lloyd
        try:
            np.random.seed(np_seed)
            labels, sizes = utility.cluster_data(X.copy(), K)
        finally:
            skm._kmeans_plusplus, skm._kmeans_single_lloyd = kpp0, lloyd0
        assert len(rec["idx"]) == 20 and len(rec["n_iter"]) == 20
        crc_final = zlib.crc32(np.ascontiguousarray(labels, dtype=np.int32).tobytes())
        winner = rec["crc"].index(crc_final)
        # sklearn's own run-to-run spread: its M-step adds per-thread partial sums in lock order
        # (_k_means_lloyd.pyx:143-152), so repeating the SAME Lloyd run from the same centres gives slightly
        # different centres, and at N ~ 7e4 other labels on a few points and sometimes another n_iter.
        # Each restart is repeated twice from its recorded seeds (KMeans(init=X[idx]) centres the init like fit does).
        from sklearn.cluster import KMeans
        rerun_n_iter = np.zeros((20, 2), dtype=np.int64)
        rerun_label_diff = np.zeros((20, 2), dtype=np.int64)
        rerun_rel_gap = np.zeros((20, 2))  # largest relative distance gap |d_a - d_b| / max(d_a, d_b) of a re-labelled point
        for r in range(20):
            for t in range(2):
                km = KMeans(n_clusters=K, init=X[rec["idx"][r]].copy(), n_init=1).fit(X)
                rerun_n_iter[r, t] = km.n_iter_
                mism = np.nonzero(km.labels_ != rec["labels"][r])[0]
                rerun_label_diff[r, t] = mism.size
                if mism.size:
                    C = km.cluster_centers_.astype(np.float64)
                    x = X[mism].astype(np.float64)
                    da = ((x - C[km.labels_[mism]]) ** 2).sum(axis=1)
                    db = ((x - C[rec["labels"][r][mism]]) ** 2).sum(axis=1)
                    rerun_rel_gap[r, t] = float((np.abs(da - db) / np.maximum(da, db)).max())
        save(name, seed=seed, n=n, d=d, k_true=k_true, K=K, spread=spread, noise=noise, np_seed=np_seed,
             x_crc=zlib.crc32(X.tobytes()), kpp_indices=np.stack(rec["idx"]), n_iter=np.asarray(rec["n_iter"]),
             inertia=np.asarray(rec["inertia"]), labels_crc=np.asarray(rec["crc"], dtype=np.int64),
             restart_sizes=np.stack(rec["sizes"]), winner=winner, rerun_n_iter=rerun_n_iter,
             rerun_label_diff=rerun_label_diff, rerun_rel_gap=rerun_rel_gap,
             labels=labels.astype(np.uint8 if K <= 256 else np.uint16), sizes=sizes)


def gen_g15(utility, only=None):
    """G14's six configurations again, with the reference stack held to ONE thread
    (``threadpool_limits(1)`` around ``np.random.seed(s); utility.cluster_data(X, K)``): sklearn's M-step then adds
    every cluster's rows in row order into one float32 buffer (_k_means_lloyd.pyx:215-218; the thread-local buffer is
    added once to zeros), so the whole fit is deterministic -- asserted here by running every restart's Lloyd loop a
    second time from its recorded seeds -- and the product's ``sums="sklearn_1t"`` mode can be compared with it bit
    for bit: per restart the k-means++ indices, n_iter, the float32 inertia, a digest and the size histogram of the
    labels; the winner; the final labels and sizes.  The fixture also records the stack it was made on (OpenBLAS
    core: the order of the seeding's float32 potentials is that of its sdot / sgemv_t kernels)."""
    import zlib
    import sklearn
    import sklearn.cluster._kmeans as skm
    from sklearn.cluster import KMeans
    from threadpoolctl import threadpool_info, threadpool_limits

    info = threadpool_info()
    blas = [i for i in info if i.get("user_api") == "blas"]
    stack = "numpy %s; scikit-learn %s; %s" % (np.__version__, sklearn.__version__, "; ".join(
        "%s %s core=%s" % (i.get("internal_api"), i.get("version"), i.get("architecture")) for i in blas))
    for name, seed, n, d, k_true, K, spread, noise, np_seed in G14_CASES:
        name = name.replace("g14_fit_", "g15_fit1t_")
        if only and name not in only:
            continue
        X = datagen.make_encodings(seed, n, d, k_true, spread, noise)
        rec = dict(idx=[], n_iter=[], inertia=[], crc=[], sizes=[])
        kpp0, lloyd0 = skm._kmeans_plusplus, skm._kmeans_single_lloyd

        def kpp(*a, **kw):
            centers, indices = kpp0(*a, **kw)
            rec["idx"].append(np.asarray(indices, dtype=np.int64).copy())
            return centers, indices

        def lloyd(*a, **kw):
            assert kw.get("n_threads", a[-1] if a else None) == 1, "sklearn is not running its Lloyd loop on one thread"
            labels, inertia, centers, n_iter = lloyd0(*a, **kw)
            rec["n_iter"].append(n_iter)
            rec["inertia"].append(np.float64(inertia))
            rec["crc"].append(zlib.crc32(np.ascontiguousarray(labels, dtype=np.int32).tobytes()))
            rec["sizes"].append(np.bincount(labels, minlength=K).astype(np.int32))
            return labels, inertia, centers, n_iter

        with threadpool_limits(limits=1):
            skm._kmeans_plusplus, skm._kmeans_single_lloyd = kpp, lloyd
            try:
                np.random.seed(np_seed)
                labels, sizes = utility.cluster_data(X.copy(), K)
            finally:
                skm._kmeans_plusplus, skm._kmeans_single_lloyd = kpp0, lloyd0
            assert len(rec["idx"]) == 20 and len(rec["n_iter"]) == 20
            crc_final = zlib.crc32(np.ascontiguousarray(labels, dtype=np.int32).tobytes())
            winner = rec["crc"].index(crc_final)
            # deterministic?  every restart's Lloyd run once more from its recorded seeds
            for r in range(20):
                km = KMeans(n_clusters=K, init=X[rec["idx"][r]].copy(), n_init=1).fit(X)
                assert km.n_iter_ == rec["n_iter"][r], (name, r, km.n_iter_, rec["n_iter"][r])
                assert zlib.crc32(np.ascontiguousarray(km.labels_, dtype=np.int32).tobytes()) == rec["crc"][r], (name, r)
                assert np.float64(km.inertia_) == rec["inertia"][r], (name, r)
        save(name, seed=seed, n=n, d=d, k_true=k_true, K=K, spread=spread, noise=noise, np_seed=np_seed,
             x_crc=zlib.crc32(X.tobytes()), kpp_indices=np.stack(rec["idx"]), n_iter=np.asarray(rec["n_iter"]),
             inertia=np.asarray(rec["inertia"]), labels_crc=np.asarray(rec["crc"], dtype=np.int64),
             restart_sizes=np.stack(rec["sizes"]), winner=winner, stack=np.array(stack),
             labels=labels.astype(np.uint8 if K <= 256 else np.uint16), sizes=sizes)


def gen_g16():
    """Fewer distinct points than clusters (300 rows drawn from 5 distinct ones, K = 8): the degenerate corner of
    sklearn's Lloyd loop -- clusters stay empty, the relocation finds (nearly) zero distances, `_average_centers`
    copies the heaviest cluster's centre (_k_means_common.pyx:291-295), KMeans warns that fewer distinct clusters
    than asked were found -- and of k-means++ (all potentials zero once every distinct point is a centre: the
    remaining draws all land on index 0).  Recorded on ONE thread (threadpool_limits(1)): in this corner sklearn's
    result depends on the rounding of its float32 sums, so its 8-thread and 1-thread runs differ in the empty
    clusters' centres; the one-thread run is the deterministic one and what sums="sklearn_1t" must reproduce."""
    from sklearn.cluster import KMeans
    from sklearn.cluster._kmeans import _kmeans_plusplus
    from sklearn.utils.extmath import row_norms
    from threadpoolctl import threadpool_limits
    rs = np.random.RandomState(3)
    base = rs.randn(5, 96).astype(np.float32)
    X = base[rs.randint(0, 5, size=300)].copy()
    K = 8
    C0 = X[:K].copy()
    out = dict(X=X, C0=C0, K=K)
    with threadpool_limits(limits=1), warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        km = KMeans(n_clusters=K, init=C0, n_init=1).fit(X)
        assert any("distinct clusters" in str(m.message) for m in w)
        out.update(labels=km.labels_.astype(np.int32), centers=km.cluster_centers_, n_iter=km.n_iter_, inertia=np.float64(km.inertia_))
        Xc = X - X.mean(axis=0)
        idx = []
        for seed in (0, 1, 2):
            _, i = _kmeans_plusplus(Xc, K, row_norms(Xc, squared=True), np.ones(300, dtype=np.float32), np.random.RandomState(seed))
            idx.append(i)
        out["kpp_indices"] = np.stack(idx).astype(np.int64)
        np.random.seed(5)
        km2 = KMeans(n_clusters=K, n_init=3).fit(X)
        out.update(fit_labels=km2.labels_.astype(np.int32), fit_centers=km2.cluster_centers_, fit_n_iter=km2.n_iter_,
                   fit_inertia=np.float64(km2.inertia_))
    save("g16_fewer_distinct_points_than_clusters", **out)


def gen_g17():
    """The ends of the cluster-count range on one thread: K = 1, 2, N - 1, N for N = 200 rows (d = 96), the whole seeded
    ``KMeans(K, n_init=3).fit`` -- k-means++ with L = 2 + int(ln K) local trials from 2 to 7, a single cluster, a run
    where every point is a centre (inertia 0, strict convergence in the first iteration)."""
    from sklearn.cluster import KMeans
    from threadpoolctl import threadpool_limits
    X = datagen.make_encodings(5, 200, 96, 6, 2000, 500)
    out = dict(X=X, Ks=np.array([1, 2, 199, 200]), np_seed=4)
    with threadpool_limits(limits=1), warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for K in (1, 2, 199, 200):
            np.random.seed(4)
            km = KMeans(n_clusters=K, n_init=3).fit(X)
            out[f"labels_{K}"] = km.labels_.astype(np.int32)
            out[f"centers_{K}"] = km.cluster_centers_
            out[f"n_iter_{K}"] = km.n_iter_
            out[f"inertia_{K}"] = np.float64(km.inertia_)
    save("g17_extreme_cluster_counts", **out)


if __name__ == "__main__":
    warnings.simplefilter("error")  # a ConvergenceWarning here means a bad fixture
    utility, metrics = _import_ref("PCCSNet")
    only = [a for a in sys.argv[1:] if not a.startswith(("g14_", "g15_"))]
    g14_only = [a for a in sys.argv[1:] if a.startswith("g14_")]  # e.g. `gen_golden.py g14 g14_fit_univ_n30128_k200`
    g15_only = [a for a in sys.argv[1:] if a.startswith("g15_")]
    jobs = dict(g1=lambda: gen_g1(utility), g2=lambda: gen_g2(utility), g45=gen_g4_g5,
                g6=lambda: gen_g6(metrics), g7=gen_g7, g8=gen_g8, g10=gen_g10, g11=gen_g11, g12=gen_g12, g13=gen_g13,
                g14=lambda: gen_g14(utility, g14_only or None), g15=lambda: gen_g15(utility, g15_only or None), g16=gen_g16, g17=gen_g17)
    for name, fn in jobs.items():
        if not only or name in only:
            fn()
