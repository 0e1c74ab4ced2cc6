"""Modality clustering on MI355X: the host side of the reference's k-means call sites.

Mirrors (same names, argument meaning and error behaviour)
    PCCSNet/utils/utility.py:64-133   get_memory_data, get_modality, cluster_data, k_means_cluster
    MomentaryPrediction/PCCSNet_XOE/utils/utility.py:67-144 (same + ``pm`` scaling, ``reC``)
and the scikit-learn driver those call, ``KMeans(n_clusters, n_init=20).fit`` (sklearn 1.7.2,
cluster/_kmeans.py:1427-1554), restated in ``kmeans_fit``.  All arithmetic on the points runs
in the HIP kernels of libmtp_accel.so; this file only sequences them.
"""
import copy
import os
import warnings

import numpy as np
import torch

from . import kernels as K
from . import seeding
from .checkings import check_cluster_consistency, check_memory_consistency


class ConvergenceWarning(UserWarning):
    """Same role as sklearn.exceptions.ConvergenceWarning (fewer distinct clusters than asked)."""


def _as_device_f32(data, device=None):
    """Accept np.ndarray / torch tensor / array-like; return a float32 device tensor (a copy)."""
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device())
    if isinstance(data, torch.Tensor):
        t = data
    else:
        if not isinstance(data, np.ndarray):
            data = np.array(data).astype(float)  # PCCSNet/utils/utility.py:109-110
        t = torch.from_numpy(np.ascontiguousarray(data))
    if t.dtype == torch.float64:
        # PCCSNet/utils/utility.py:109-110 turns a non-ndarray into float64 and sklearn then clusters in float64.
        # The kernels here are float32 (what the reference's own call sites pass: memory.npy is float32); casting
        # silently would return labels that are not the reference's, so the input is refused.
        raise TypeError("pccs_accel clusters float32 encodings (what the reference's call sites pass); got float64 -- "
                        "cast with .astype(np.float32) if float32 clustering is what you want")
    if t.dim() != 2:
        raise ValueError(f"Expected 2D array, got shape {tuple(t.shape)}")
    return t.to(device=device, dtype=torch.float32, copy=True).contiguous()


def tolerance_and_mean(X, tol, order="numpy"):
    """(tol_abs, mean32) as KMeans.fit computes them before centring (sklearn cluster/_kmeans.py:279-287,
    :1478-1481): ``X.mean(axis=0)`` and ``np.mean(np.var(X, axis=0)) * tol``.

    ``order="numpy"``: the column sums are taken in NumPy's own float32 order (row after row,
    ``mtp_col_seqsum_f32``) and divided by NumPy on the host exactly as numpy/_core/_methods.py does, so the
    mean, the centred points and the tolerance are the reference's bit for bit.  ``order="f64"``: double sums
    (parallel, independent of how the rows are split; what the sharded fit uses): the mean can differ by an
    ulp, which moves a few boundary points per 7e4."""
    n, d = X.shape
    if order == "numpy":
        s = K.col_seqsum(X).cpu().numpy()
        mean = np.true_divide(s, np.intp(n), out=s, casting="unsafe")  # _mean: ret / rcount into the float32 array
        mean32 = torch.from_numpy(mean).to(X.device)
        if tol == 0:
            return 0.0, mean32
        sq = K.col_seqsum(X, mean32).cpu().numpy()
        var = np.true_divide(sq, np.intp(n), out=sq, casting="unsafe")  # _var
        return float(np.mean(var) * tol), mean32
    if order != "f64":
        raise ValueError(f"order must be 'numpy' or 'f64', got {order!r}")
    col_sum = K.col_sums(X)
    mean64 = col_sum / n
    sqdev = K.col_sqdev(X, mean64)
    tol_abs = float((sqdev / n).mean().item()) * tol if tol != 0 else 0.0
    return tol_abs, mean64.to(torch.float32)


def is_same_clustering(l1, l2, k):
    """sklearn cluster/_k_means_common.pyx:314-328: l1 -> l2 is a function (device check).  Every label of l1
    remembers the l2 label of ONE of its points (which one does not matter); the mapping is a function iff every
    point agrees with what its label remembers.  Two elementwise launches and one host read."""
    i1 = l1.to(torch.int64)
    mapping = torch.full((int(k),), -1, dtype=l2.dtype, device=l1.device)
    mapping[i1] = l2
    return bool((mapping[i1] == l2).all())


def restart_batch(n, d, k, n_init):
    """How many restarts share one launch.  Small problems (one Lloyd iteration is launch-bound
    and cannot fill 256 CUs) run all restarts together; big ones gain nothing from it."""
    if d not in (64, 96, 128):
        return 1
    return int(max(1, min(n_init, 4e10 // (float(n) * k * d))))


def kmeans_fit(X, n_clusters, n_init=20, init="k-means++", random_state=None, max_iter=300, tol=1e-4,
               x_mean=None, check_every=None, batch=None, pot_order="openblas", restarts_out=None, stats_order="numpy",
               sums="f64"):
    """sklearn ``KMeans(n_clusters, n_init=n_init, init=init).fit(X)`` on one MI355X.

    X: float32 device tensor [N, D] (not modified).  ``init``: 'k-means++' or a [K, D] array /
    tensor of initial centres (then one run, like sklearn).  ``random_state``: None (NumPy's
    global RandomState, so ``np.random.seed(s)`` before the call pins it, as with the
    reference), an int, or a RandomState.  ``x_mean`` overrides the column mean (tests).
    ``batch``: restarts per launch (default: ``restart_batch``); the result does not depend on it.
    ``stats_order``: float order of the column mean / variance (``tolerance_and_mean``).
    ``pot_order``: summation order of the k-means++ potentials (``seeding``: "openblas" reproduces the
    reference stack's seeds, "f64" sums in double).  ``sums``: "f64" (default) -- double centroid sums, independent
    of the workgroup schedule and of the GPU count, within ~1e-6 of any sklearn run (whose float32 sums depend on
    its thread scheduling); "sklearn_1t" -- the float32 orders of sklearn on ONE thread (row-order chains per
    cluster and feature, NumPy's pairwise total shift, float32 inertia chain): together with the defaults of
    ``stats_order`` and ``pot_order`` the whole fit is then sklearn's under ``threadpool_limits(1)`` bit for bit
    (fixture G15), at the price of a slower M-step.  ``restarts_out``: a list that receives one
    dict(indices, inertia, n_iter, labels) per restart, in order (tests / diagnostics).
    Returns dict(labels int32 [N] (device), centers float32 [K, D] (device), inertia, n_iter, tol_abs,
    x_mean float32 [D] (device): the column mean the points were centred with).
    """
    if not (isinstance(X, torch.Tensor) and X.is_cuda and X.dtype == torch.float32 and X.dim() == 2):
        raise TypeError("kmeans_fit expects a float32 HIP device tensor [N, D]")
    n, d = X.shape
    if n < n_clusters:
        raise ValueError(f"n_samples={n} should be >= n_clusters={n_clusters}.")
    if not bool(torch.isfinite(X).all()):  # sklearn's check_array (cluster/_kmeans.py:1463) refuses these too
        raise ValueError("Input X contains NaN or infinity.")
    if random_state is None:
        rs = np.random.mtrand._rand
    elif isinstance(random_state, (int, np.integer)):
        rs = np.random.RandomState(random_state)
    else:
        rs = random_state

    X = X.contiguous()
    tol_abs, mean32 = tolerance_and_mean(X, tol, stats_order)
    if x_mean is not None:
        mean32 = torch.as_tensor(x_mean, dtype=torch.float32, device=X.device).contiguous()
    Xc = X.clone()
    K.sub_rowvec_(Xc, mean32)

    init_is_array = not isinstance(init, str)
    if init_is_array:
        c_init = torch.as_tensor(init, device=X.device).to(torch.float32).clone().contiguous()
        if tuple(c_init.shape) != (n_clusters, d):
            raise ValueError(f"The shape of the initial centers {tuple(c_init.shape)} does not match "
                             f"the number of clusters {n_clusters} / features {d}.")
        if n_init != 1:
            warnings.warn("Explicit initial center position passed: performing only one init", RuntimeWarning,
                          stacklevel=2)
        n_init = 1
        K.sub_rowvec_(c_init, mean32)
    elif init != "k-means++":
        raise ValueError(f"init must be 'k-means++' or an array, got {init!r}")

    best = None

    def consider(labels, inertia, centers, n_iter, indices=None):
        # sklearn cluster/_kmeans.py:1525-1532, restarts visited in order
        nonlocal best
        if restarts_out is not None:
            restarts_out.append(dict(indices=indices, inertia=inertia, n_iter=n_iter, labels=labels))
        if best is None or (inertia < best["inertia"] and not is_same_clustering(labels, best["labels"], n_clusters)):
            best = dict(labels=labels, centers=centers, inertia=inertia, n_iter=n_iter)

    if batch is None:
        batch = restart_batch(n, d, n_clusters, n_init)
    if init_is_array:
        consider(*K.lloyd(Xc, c_init, max_iter=max_iter, tol_abs=tol_abs, check_every=check_every, sums=sums))
    elif batch <= 1:
        # Lloyd one restart at a time, but the restarts are still SEEDED in groups: with one restart
        # half of the seeding kernel's 16-row candidate tile is empty.  Same seeds as one by one
        # (the draws are made up front in sklearn's order; seeding consumes the stream, Lloyd does not).
        draws = [seeding.draw(rs, n, n_clusters) for _ in range(n_init)]
        l = 2 + int(np.log(n_clusters))
        group = int(max(1, min(n_init, 16e9 // (2.0 * l * n * 4))))  # closest-distance rows: 2 * L * N floats each
        for s0 in range(0, n_init, group):
            c0s, idx = seeding.kmeans_plusplus_batched(Xc, n_clusters, draws[s0:s0 + group], pot_order=pot_order)
            for b in range(c0s.shape[0]):  # a copy: row b of the batch starts 16-byte aligned only when 4 | K * D
                consider(*K.lloyd(Xc, c0s[b].clone(), max_iter=max_iter, tol_abs=tol_abs, check_every=check_every,
                                  sums=sums), indices=idx[b])
            del c0s
    else:
        # the draws of every restart, in sklearn's order (seeding consumes the stream, Lloyd does not)
        draws = [seeding.draw(rs, n, n_clusters) for _ in range(n_init)]
        for s0 in range(0, n_init, batch):
            c0s, idx = seeding.kmeans_plusplus_batched(Xc, n_clusters, draws[s0:s0 + batch], pot_order=pot_order)
            labels, inertia, centers, n_iter = K.lloyd_batched(Xc, c0s, max_iter=max_iter, tol_abs=tol_abs,
                                                               check_every=check_every or 4, sums=sums)
            inertia = inertia.cpu().numpy()
            for b in range(c0s.shape[0]):
                consider(labels[b], float(inertia[b]), centers[b], n_iter[b], indices=idx[b])
        best["labels"] = best["labels"].clone()
        best["centers"] = best["centers"].clone()

    best["centers"] = best["centers"] + mean32
    distinct = int(torch.unique(best["labels"]).numel())
    if distinct < n_clusters:
        warnings.warn(f"Number of distinct clusters ({distinct}) found smaller than n_clusters ({n_clusters}). "
                      "Possibly due to duplicate points in X.", ConvergenceWarning, stacklevel=2)
    best["tol_abs"] = tol_abs
    best["x_mean"] = mean32
    return best


# ------------------------------------------------------------------------------------------
# The reference's call sites
# ------------------------------------------------------------------------------------------
# Summation mode of the reference-named call sites below (their signatures are the reference's and have no room
# for it): "f64" (``kmeans_fit``'s default) or "sklearn_1t" -- the reference stack on one thread, bit for bit.
# Set the attribute, or PCCS_ACCEL_SUMS in the environment before the import.
SUMS = os.environ.get("PCCS_ACCEL_SUMS", "f64")


def k_means_cluster(data, n_clusters, init=20):
    """PCCSNet/utils/utility.py:120-133.  ``init`` is sklearn's n_init.

    Returns (cluster_result np.int32 [N], size_array np.int64 [K]); raises KeyError(i) when
    cluster i ends up empty, as the reference's ``res[i]`` lookup does."""
    if SUMS not in ("f64", "sklearn_1t"):
        raise ValueError(f"pccs_accel.clustering.SUMS must be 'f64' or 'sklearn_1t', got {SUMS!r}")
    X = _as_device_f32(data)
    fit = kmeans_fit(X, n_clusters, n_init=init, sums=SUMS)
    labels = fit["labels"]
    sizes = torch.bincount(labels.to(torch.int64), minlength=n_clusters).cpu().numpy().astype(np.int64)
    for i in range(n_clusters):
        if sizes[i] == 0:
            raise KeyError(i)
    return labels.cpu().numpy(), sizes


def cluster_data(data, n_cluster=200, save_dir=None, args=None):
    """PCCSNet/utils/utility.py:108-117; with ``args`` the XOE variant
    (PCCSNet_XOE/utils/utility.py:115-128): ``data[:, :args.obs_feat_size] *= args.pm`` first."""
    if not isinstance(data, (np.ndarray, torch.Tensor)):
        data = np.array(data).astype(float)
    if args is not None:
        print("pm %.4f Applied" % args.pm)
        data[:, :args.obs_feat_size] *= args.pm
    cluster_result, size_array = k_means_cluster(data, n_cluster)
    if save_dir is not None:
        np.save(os.path.join(save_dir, "cluster_result.npy"), cluster_result)
        np.save(os.path.join(save_dir, "size_array.npy"), size_array)
    return cluster_result, size_array


def get_memory_data(dataset, save_dir, args):
    """PCCSNet/utils/utility.py:64-85 (and the XOE copy, :67-92, when ``args`` has reC / pm)."""
    mem_path = os.path.join(save_dir, "memory.npy")
    if os.path.exists(mem_path) and check_memory_consistency(save_dir, args):
        memory_data = np.load(mem_path)
    else:
        print("==> Generating Memory...")
        memory_data = torch.cat([dataset.obs_enc, dataset.pred_enc], dim=1)
        memory_data = memory_data.cpu().numpy()
        np.save(mem_path, memory_data)

    print("==> Memory Shape", memory_data.shape)

    xoe = hasattr(args, "reC")
    try:
        cluster_result = np.load(os.path.join(save_dir, "cluster_result.npy"))
        size_array = np.load(os.path.join(save_dir, "size_array.npy"))
        if not check_cluster_consistency(save_dir, size_array, args):
            raise FileNotFoundError
        if xoe and args.reC:
            raise FileNotFoundError
    except FileNotFoundError:
        if xoe:
            args.reC = False
        print("==> Clustering Memory...")
        cluster_result, size_array = cluster_data(copy.deepcopy(memory_data), n_cluster=args.n_cluster,
                                                  save_dir=save_dir, args=args if xoe else None)
    return memory_data, cluster_result, size_array


def classifier_targets(cluster_result, n_cluster, device=None, one_hot=True):
    """The label hand-off of PCCSNet/train_steps.py:357 and :364-366 in one launch.

    ``cluster_result``: the int labels ``get_memory_data`` / ``cluster_data`` return (NumPy array or tensor).
    Returns ``(cluster_results, gt_prob)``: ``torch.from_numpy(cluster_result).long()`` -- what the reference stores
    in ``train_set.cluster_results`` -- and, when ``one_hot`` (the reference's ``--disable_modality_loss`` branch),
    ``gt_prob`` float32 [N, n_cluster] with ``gt_prob[i, cluster_result[i]] = 1`` (else None), both on the device.
    Out-of-range labels raise IndexError like the reference's fancy index."""
    if isinstance(cluster_result, torch.Tensor):
        lab = cluster_result
    else:
        lab = torch.from_numpy(np.ascontiguousarray(cluster_result))
    if lab.dim() != 1:
        raise ValueError(f"Expected a 1D label array, got shape {tuple(lab.shape)}")
    if device is None:
        device = lab.device if lab.is_cuda else torch.device("cuda", torch.cuda.current_device())
    lab = lab.to(device=device, dtype=torch.int32).contiguous()
    return K.label_targets(lab, int(n_cluster), want_long=True, want_one_hot=one_hot)


def get_modality(memory_data, cluster_result, num_cluster):
    """PCCSNet/utils/utility.py:98-105: per-cluster mean of the member rows -> float32 [K, D].

    ``memory_data`` may live on the CPU (as at train_steps.py:356) or on the GPU; the result is
    returned on the input's device so the caller's ``.cuda()`` keeps working.  Raises
    KeyError(i) for an empty cluster like the reference's dict lookup."""
    src_device = memory_data.device if isinstance(memory_data, torch.Tensor) else torch.device("cpu")
    X = _as_device_f32(memory_data)
    labels = torch.as_tensor(np.asarray(cluster_result) if not isinstance(cluster_result, torch.Tensor)
                             else cluster_result).to(device=X.device, dtype=torch.int32).contiguous()
    if labels.numel() != X.shape[0]:
        raise ValueError("cluster_result and memory_data disagree on the number of samples")
    # labels outside [0, num_cluster) are ignored, as the reference's dict build ignores them
    sums, counts = K.centroid_sums(X, labels, num_cluster)
    means, n_empty = K.cluster_means(sums, counts)
    if int(n_empty.item()) > 0:
        raise KeyError(int(torch.nonzero(counts == 0)[0].item()))
    return means.to(src_device)
