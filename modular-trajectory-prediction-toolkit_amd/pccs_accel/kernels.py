"""Torch-tensor front ends of the C-ABI entry points (one function per entry point).

PyTorch is plumbing here: it owns device memory and the stream; every operation below is a
hand-written HIP kernel reached through ctypes.  Shapes, dtypes, devices and contiguity are
checked on the host before a kernel sees a pointer.
"""
import ctypes

import torch

from . import _lib
from ._lib import check, lib, ptr, stream_ptr


def _req(t, name, dtype, ndim=None, align=16):
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name}: expected a torch.Tensor, got {type(t).__name__}")
    if not t.is_cuda:
        raise _lib.MtpError(f"{name}: pccs_accel kernels need a HIP device tensor (got device={t.device}); "
                            "there is no CPU path")
    if t.dtype != dtype:
        raise TypeError(f"{name}: expected dtype {dtype}, got {t.dtype}")
    if ndim is not None and t.dim() != ndim:
        raise ValueError(f"{name}: expected {ndim} dimensions, got shape {tuple(t.shape)}")
    if not t.is_contiguous():
        raise ValueError(f"{name}: must be contiguous")
    if t.numel() > 0 and t.data_ptr() % align != 0:
        raise ValueError(f"{name}: storage must be {align}-byte aligned")
    return t


def staged_centers_floats(k, d):
    return int(lib().mtp_staged_centers_floats(k, d))


def row_sqnorm(A):
    _req(A, "A", torch.float32, 2)
    out = torch.empty(A.shape[0], dtype=torch.float32, device=A.device)
    with torch.cuda.device(A.device):
        check(lib().mtp_row_sqnorm_f32(ptr(A), A.shape[0], A.shape[1], ptr(out), stream_ptr()), "mtp_row_sqnorm_f32")
    return out


def col_sums(X, out=None):
    _req(X, "X", torch.float32, 2)
    if out is None:
        out = torch.zeros(X.shape[1], dtype=torch.float64, device=X.device)
    _req(out, "out", torch.float64, 1)
    with torch.cuda.device(X.device):
        check(lib().mtp_col_sums_f32(ptr(X), X.shape[0], X.shape[1], ptr(out), stream_ptr()), "mtp_col_sums_f32")
    return out


def col_sqdev(X, mean64, out=None):
    _req(X, "X", torch.float32, 2)
    _req(mean64, "mean", torch.float64, 1)
    assert mean64.numel() == X.shape[1]
    if out is None:
        out = torch.zeros(X.shape[1], dtype=torch.float64, device=X.device)
    with torch.cuda.device(X.device):
        check(lib().mtp_col_sqdev_f32(ptr(X), X.shape[0], X.shape[1], ptr(mean64), ptr(out), stream_ptr()),
              "mtp_col_sqdev_f32")
    return out


def col_seqsum(X, center=None, init=None):
    """float32 [d]: the columns of X (or of (X - center)**2, float arithmetic) summed row after row in
    float32 -- NumPy's ``np.add.reduce(X, axis=0)`` order (``mtp_col_seqsum_carry_f32``).  ``init``: the sums
    of the rows before these (row-sharded points), else zero."""
    _req(X, "X", torch.float32, 2)
    for name, t in (("center", center), ("init", init)):
        if t is not None:
            _req(t, name, torch.float32, 1)
            assert t.numel() == X.shape[1]
    out = torch.empty(X.shape[1], dtype=torch.float32, device=X.device)
    with torch.cuda.device(X.device):
        check(lib().mtp_col_seqsum_carry_f32(ptr(X), X.shape[0], X.shape[1], ptr(center), ptr(init), ptr(out),
                                             stream_ptr()), "mtp_col_seqsum_carry_f32")
    return out


def sub_rowvec_(X, vec):
    _req(X, "X", torch.float32, 2)
    _req(vec, "vec", torch.float32, 1)
    assert vec.numel() == X.shape[1]
    with torch.cuda.device(X.device):
        check(lib().mtp_sub_rowvec_f32(ptr(X), X.shape[0], X.shape[1], ptr(vec), stream_ptr()), "mtp_sub_rowvec_f32")
    return X


def assign(X, C, labels=None, n_changed=None):
    """E-step.  labels (int32 [N]) is updated in place when given (previous labels in, new
    labels out) and returned.  n_changed: optional float64 [1] tensor, incremented."""
    _req(X, "X", torch.float32, 2)
    _req(C, "C", torch.float32, 2)
    n, d = X.shape
    k = C.shape[0]
    if C.shape[1] != d:
        raise ValueError(f"C has {C.shape[1]} features, X has {d}")
    if labels is None:
        labels = torch.full((n,), -1, dtype=torch.int32, device=X.device)
    _req(labels, "labels", torch.int32, 1, align=4)
    assert labels.numel() == n
    if n_changed is not None:
        _req(n_changed, "n_changed", torch.float64, 1)
    ws_bytes = int(lib().mtp_assign_workspace_bytes(k, d))
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=X.device)
    with torch.cuda.device(X.device):
        check(lib().mtp_assign_f32(ptr(X), n, d, ptr(C), k, None, ptr(labels), ptr(n_changed), ptr(ws), ws_bytes,
                                   stream_ptr()), "mtp_assign_f32")
    return labels


def centroid_sums(X, labels, k, sums=None, counts=None):
    """sums [k, d] float64 += member rows, counts [k] float64 += member counts."""
    _req(X, "X", torch.float32, 2)
    _req(labels, "labels", torch.int32, 1, align=4)
    n, d = X.shape
    assert labels.numel() == n
    if sums is None:
        sums = torch.zeros((k, d), dtype=torch.float64, device=X.device)
    if counts is None:
        counts = torch.zeros(k, dtype=torch.float64, device=X.device)
    _req(sums, "sums", torch.float64, 2)
    _req(counts, "counts", torch.float64, 1)
    assert tuple(sums.shape) == (k, d) and counts.numel() == k
    with torch.cuda.device(X.device):
        check(lib().mtp_centroid_sums_f32(ptr(X), n, d, ptr(labels), k, ptr(sums), ptr(counts), stream_ptr()),
              "mtp_centroid_sums_f32")
    return sums, counts


def cluster_means(sums, counts):
    _req(sums, "sums", torch.float64, 2)
    _req(counts, "counts", torch.float64, 1)
    k, d = sums.shape
    means = torch.zeros((k, d), dtype=torch.float32, device=sums.device)
    n_empty = torch.zeros(1, dtype=torch.int32, device=sums.device)
    with torch.cuda.device(sums.device):
        check(lib().mtp_cluster_means_f32(ptr(sums), ptr(counts), k, d, ptr(means), ptr(n_empty), stream_ptr()),
              "mtp_cluster_means_f32")
    return means, n_empty


def label_targets(labels, k, want_long=True, want_one_hot=True):
    """(labels int64 [N] or None, one-hot float32 [N, K] or None) on labels' device, for int32 device labels in [0, K)."""
    _req(labels, "labels", torch.int32, 1, align=4)
    n = labels.shape[0]
    if n and (int(labels.min()) < 0 or int(labels.max()) >= k):  # the reference's fancy index raises IndexError here
        raise IndexError(f"cluster label out of range for {k} clusters")
    dev = labels.device
    lab64 = torch.empty(n, dtype=torch.int64, device=dev) if want_long else None
    hot = torch.empty((n, k), dtype=torch.float32, device=dev) if want_one_hot else None
    with torch.cuda.device(dev):
        check(lib().mtp_label_targets_f32(ptr(labels), n, k, ptr(lab64), ptr(hot), stream_ptr()), "mtp_label_targets_f32")
    return lab64, hot


def inertia(X, C, labels, per_point=False):
    _req(X, "X", torch.float32, 2)
    _req(C, "C", torch.float32, 2)
    _req(labels, "labels", torch.int32, 1, align=4)
    n, d = X.shape
    assert C.shape[1] == d and labels.numel() == n
    if n and (int(labels.min()) < 0 or int(labels.max()) >= C.shape[0]):  # the kernel indexes C with these
        raise IndexError(f"cluster label out of range for {C.shape[0]} centres")
    out = torch.zeros(1, dtype=torch.float64, device=X.device)
    pp = torch.empty(n, dtype=torch.float32, device=X.device) if per_point else None
    with torch.cuda.device(X.device):
        check(lib().mtp_inertia_f32(ptr(X), n, d, ptr(C), ptr(labels), ptr(out), ptr(pp), stream_ptr()),
              "mtp_inertia_f32")
    return (out, pp) if per_point else out


def _sums_option(sums):
    """``sums``: "f64" (default: double centroid sums, schedule-independent) or "sklearn_1t" (float32 row-order
    chains, NumPy-order total shift, float32 inertia chain: what sklearn computes with one thread, bit for bit)."""
    if sums in (None, "f64"):
        return 0
    if sums == "sklearn_1t":
        return _lib.LLOYD_SUMS_SKLEARN_1T
    raise ValueError(f"sums must be 'f64' or 'sklearn_1t', got {sums!r}")


def lloyd(Xc, C0, max_iter=300, tol_abs=0.0, check_every=None, sums="f64"):
    """Single-GPU sklearn ``_kmeans_single_lloyd`` on centred data.  ``sums``: see ``_sums_option``.

    Returns (labels int32 [N], inertia float, centers float32 [K, D], n_iter)."""
    _req(Xc, "X", torch.float32, 2)
    _req(C0, "C0", torch.float32, 2)
    n, d = Xc.shape
    k = C0.shape[0]
    if C0.shape[1] != d:
        raise ValueError(f"C0 has {C0.shape[1]} features, X has {d}")
    if n < k:
        raise ValueError(f"n_samples={n} should be >= n_clusters={k}.")
    if check_every is None:
        # sync with the host rarely when an iteration is short, every iteration when it is long
        check_every = 1 if float(n) * k * d > 2e10 else 8
    C = C0.clone()
    labels = torch.empty(n, dtype=torch.int32, device=Xc.device)
    inert = torch.zeros(1, dtype=torch.float64, device=Xc.device)
    ws_bytes = int(lib().mtp_lloyd_workspace_bytes(k, d)) + int(lib().mtp_far_points_scratch_bytes(n))
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=Xc.device)
    n_iter = ctypes.c_int(0)
    with torch.cuda.device(Xc.device):
        check(lib().mtp_lloyd_f32(ptr(Xc), n, d, ptr(C), k, int(max_iter), float(tol_abs), int(check_every),
                                  ptr(labels), ptr(inert), ctypes.byref(n_iter), _sums_option(sums), ptr(ws), ws_bytes,
                                  stream_ptr()),
              "mtp_lloyd_f32")
    return labels, float(inert.item()), C, int(n_iter.value)


def lloyd_batched(Xc, C0s, max_iter=300, tol_abs=0.0, check_every=4, sums="f64"):
    """``C0s.shape[0]`` independent ``lloyd`` runs on the same centred points, sharing every launch
    (the n_init restarts of KMeans.fit).  C0s [B, K, D].  Returns (labels int32 [B, N],
    inertia float64 [B] (device), centers float32 [B, K, D], n_iter list of B ints)."""
    _req(Xc, "X", torch.float32, 2)
    _req(C0s, "C0s", torch.float32, 3)
    n, d = Xc.shape
    b, k, d2 = C0s.shape
    if d2 != d:
        raise ValueError(f"C0s has {d2} features, X has {d}")
    if n < k:
        raise ValueError(f"n_samples={n} should be >= n_clusters={k}.")
    C = C0s.clone()
    labels = torch.empty((b, n), dtype=torch.int32, device=Xc.device)
    inert = torch.zeros(b, dtype=torch.float64, device=Xc.device)
    ws_bytes = int(lib().mtp_lloyd_batched_workspace_bytes(k, d, b, n))
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=Xc.device)
    n_iter = (ctypes.c_int * b)()
    with torch.cuda.device(Xc.device):
        check(lib().mtp_lloyd_batched_f32(ptr(Xc), n, d, ptr(C), k, b, int(max_iter), float(tol_abs), int(check_every),
                                          ptr(labels), ptr(inert), n_iter, _sums_option(sums), ptr(ws), ws_bytes,
                                          stream_ptr()),
              "mtp_lloyd_batched_f32")
    return labels, inert, C, [int(v) for v in n_iter]


def best_of_k(pred, gt, mode=0, ade_div=None, best_of=None, want_per_agent=True):
    """pred [N, Ktot, T, 2], gt [N, T, 2] (float32 or float64, same dtype).

    Returns dict(min_ade, min_fde, idx, idx2 (per agent, or None), sums float64 [2] on device)."""
    if pred.dtype not in (torch.float32, torch.float64):
        raise TypeError(f"pred: expected float32 or float64, got {pred.dtype}")
    _req(pred, "pred", pred.dtype, 4, align=pred.element_size())
    _req(gt, "gt", pred.dtype, 3, align=pred.element_size())
    n, ktot, t, two = pred.shape
    if two != 2 or tuple(gt.shape) != (n, t, 2):
        raise ValueError(f"shape mismatch: pred {tuple(pred.shape)}, gt {tuple(gt.shape)}")
    k = ktot if best_of is None else int(best_of)
    if not (1 <= k <= ktot):
        raise ValueError(f"best_of={k} must be in [1, {ktot}]")
    if mode not in (0, 1, 2):
        raise ValueError(f"mode must be 0, 1 or 2 (got {mode})")
    if ade_div is None:
        ade_div = t
    dev = pred.device
    sums = torch.zeros(2, dtype=torch.float64, device=dev)
    if want_per_agent:
        min_ade = torch.empty(n, dtype=pred.dtype, device=dev)
        min_fde = torch.empty(n, dtype=pred.dtype, device=dev)
        idx = torch.empty(n, dtype=torch.int32, device=dev)
        idx2 = torch.empty(n, dtype=torch.int32, device=dev)
    else:
        min_ade = min_fde = idx = idx2 = None
    fn = lib().mtp_best_of_k_f32 if pred.dtype == torch.float32 else lib().mtp_best_of_k_f64
    with torch.cuda.device(dev):
        check(fn(ptr(pred), ptr(gt), n, k, ktot, t, float(ade_div), int(mode), ptr(min_ade), ptr(min_fde), ptr(idx),
                 ptr(idx2), ptr(sums), stream_ptr()), "mtp_best_of_k")
    return dict(min_ade=min_ade, min_fde=min_fde, idx=idx, idx2=idx2, sums=sums)


def ade_fde(pred, gt, ade_div=1.0, best_of=None):
    """Per-candidate (ADE, FDE) [N, K] for float32 pred [N, Ktot, T, 2], gt [N, T, 2]."""
    _req(pred, "pred", torch.float32, 4, align=4)
    _req(gt, "gt", torch.float32, 3, align=4)
    n, ktot, t, two = pred.shape
    if two != 2 or tuple(gt.shape) != (n, t, 2):
        raise ValueError(f"shape mismatch: pred {tuple(pred.shape)}, gt {tuple(gt.shape)}")
    k = ktot if best_of is None else int(best_of)
    ade = torch.empty((n, k), dtype=torch.float32, device=pred.device)
    fde = torch.empty((n, k), dtype=torch.float32, device=pred.device)
    with torch.cuda.device(pred.device):
        check(lib().mtp_ade_fde_f32(ptr(pred), ptr(gt), n, k, ktot, t, float(ade_div), ptr(ade), ptr(fde),
                                    stream_ptr()), "mtp_ade_fde_f32")
    return ade, fde


# ------------------------------------------------------------------------------------------
# decode glue (f3) and NMS re-ranking (f4)
# ------------------------------------------------------------------------------------------
def decode_trajectories(disp, rot=None, last_obs=None, out=None):
    """Prefix sum over time of float32 displacements [N, K, T, 2] (+ rotate back with rot
    [N, 2, 2] + add last_obs [N, 2]).  ``out=disp`` decodes in place."""
    _req(disp, "disp", torch.float32, 4, align=4)
    n, k, t, two = disp.shape
    if two != 2:
        raise ValueError(f"disp: expected [N, K, T, 2], got {tuple(disp.shape)}")
    if rot is not None:
        _req(rot, "rot", torch.float32, 3)
        if tuple(rot.shape) != (n, 2, 2):
            raise ValueError(f"rot: expected {(n, 2, 2)}, got {tuple(rot.shape)}")
    if last_obs is not None:
        _req(last_obs, "last_obs", torch.float32, 2, align=4)
        if tuple(last_obs.shape) != (n, 2):
            raise ValueError(f"last_obs: expected {(n, 2)}, got {tuple(last_obs.shape)}")
    if out is None:
        out = torch.empty_like(disp)
    _req(out, "out", torch.float32, 4, align=4)
    assert out.shape == disp.shape
    with torch.cuda.device(disp.device):
        check(lib().mtp_decode_trajectories_f32(ptr(disp), n, k, t, ptr(rot), ptr(last_obs), ptr(out), stream_ptr()),
              "mtp_decode_trajectories_f32")
    return out


def best_of_k_disp(disp, gt, mode=0, ade_div=None, want_per_agent=True):
    """best_of_k on raw per-step displacements [N, K, T, 2] (float32): transform_trajectory is
    fused into the scoring pass."""
    _req(disp, "disp", torch.float32, 4, align=4)
    _req(gt, "gt", torch.float32, 3, align=4)
    n, k, t, two = disp.shape
    if two != 2 or tuple(gt.shape) != (n, t, 2):
        raise ValueError(f"shape mismatch: disp {tuple(disp.shape)}, gt {tuple(gt.shape)}")
    if mode not in (0, 1, 2):
        raise ValueError(f"mode must be 0, 1 or 2 (got {mode})")
    if ade_div is None:
        ade_div = t
    dev = disp.device
    sums = torch.zeros(2, dtype=torch.float64, device=dev)
    if want_per_agent:
        min_ade = torch.empty(n, dtype=torch.float32, device=dev)
        min_fde = torch.empty(n, dtype=torch.float32, device=dev)
        idx = torch.empty(n, dtype=torch.int32, device=dev)
        idx2 = torch.empty(n, dtype=torch.int32, device=dev)
    else:
        min_ade = min_fde = idx = idx2 = None
    with torch.cuda.device(dev):
        check(lib().mtp_best_of_k_disp_f32(ptr(disp), ptr(gt), n, k, k, t, float(ade_div), int(mode), ptr(min_ade),
                                           ptr(min_fde), ptr(idx), ptr(idx2), ptr(sums), stream_ptr()),
              "mtp_best_of_k_disp_f32")
    return dict(min_ade=min_ade, min_fde=min_fde, idx=idx, idx2=idx2, sums=sums)


def nms_rerank(pred, scores, max_samples, best_of, bin_size, nms_thresh):
    """order int32 [N, max_samples], n_selected int32 [N] (device) for pred [N, Mtot, T, 2]
    (float32 / float64) and scores [N, Mtot] (float32 / float64)."""
    if pred.dtype not in (torch.float32, torch.float64):
        raise TypeError(f"pred: expected float32 or float64, got {pred.dtype}")
    if scores.dtype not in (torch.float32, torch.float64):
        raise TypeError(f"scores: expected float32 or float64, got {scores.dtype}")
    _req(pred, "pred", pred.dtype, 4, align=pred.element_size())
    _req(scores, "scores", scores.dtype, 2, align=scores.element_size())
    n, mtot, t, two = pred.shape
    if two != 2 or tuple(scores.shape) != (n, mtot):
        raise ValueError(f"shape mismatch: pred {tuple(pred.shape)}, scores {tuple(scores.shape)}")
    if not (1 <= max_samples <= min(mtot, 1024)):
        raise ValueError(f"max_samples={max_samples} must be in [1, min({mtot}, 1024)]")
    order = torch.empty((n, max_samples), dtype=torch.int32, device=pred.device)
    n_sel = torch.empty(n, dtype=torch.int32, device=pred.device)
    fn = lib().mtp_nms_rerank_f32 if pred.dtype == torch.float32 else lib().mtp_nms_rerank_f64
    with torch.cuda.device(pred.device):
        check(fn(ptr(pred), ptr(scores), int(scores.dtype == torch.float64), n, mtot, t, int(max_samples), int(best_of),
                 float(bin_size), float(nms_thresh), ptr(order), ptr(n_sel), stream_ptr()), "mtp_nms_rerank")
    return order, n_sel


def gather_candidates(pred, order):
    """Re-ordered copy: out[a, p] = pred[a, order[a, p]] for p < order.shape[1], rest unchanged."""
    _req(pred, "pred", pred.dtype, 4)
    _req(order, "order", torch.int32, 2, align=4)
    n, mtot, t, _ = pred.shape
    m = order.shape[1]
    if order.shape[0] != n or m > mtot:
        raise ValueError(f"order {tuple(order.shape)} does not fit pred {tuple(pred.shape)}")
    if n and (int(order.min()) < 0 or int(order.max()) >= mtot):
        raise IndexError(f"order holds candidate indices outside [0, {mtot})")
    out = torch.empty_like(pred)
    fn = lib().mtp_gather_candidates_f32 if pred.dtype == torch.float32 else lib().mtp_gather_candidates_f64
    with torch.cuda.device(pred.device):
        check(fn(ptr(pred), ptr(order), n, mtot, m, t, ptr(out), stream_ptr()), "mtp_gather_candidates")
    return out


def best_of_k_reranked(pred, gt, order, mode=0, ade_div=None, best_of=None):
    """best_of_k over the first ``best_of`` positions of the re-ranked list given by ``order``."""
    if pred.dtype not in (torch.float32, torch.float64):
        raise TypeError(f"pred: expected float32 or float64, got {pred.dtype}")
    _req(pred, "pred", pred.dtype, 4, align=pred.element_size())
    _req(gt, "gt", pred.dtype, 3, align=pred.element_size())
    _req(order, "order", torch.int32, 2, align=4)
    n, ktot, t, two = pred.shape
    if two != 2 or tuple(gt.shape) != (n, t, 2) or order.shape[0] != n:
        raise ValueError(f"shape mismatch: pred {tuple(pred.shape)}, gt {tuple(gt.shape)}, order {tuple(order.shape)}")
    k = order.shape[1] if best_of is None else int(best_of)
    if not (1 <= k <= order.shape[1]):
        raise ValueError(f"best_of={k} must be in [1, {order.shape[1]}]")
    if n and (int(order.min()) < 0 or int(order.max()) >= ktot):  # the kernel indexes pred with these
        raise IndexError(f"order holds candidate indices outside [0, {ktot})")
    if ade_div is None:
        ade_div = t
    dev = pred.device
    sums = torch.zeros(2, dtype=torch.float64, device=dev)
    min_ade = torch.empty(n, dtype=pred.dtype, device=dev)
    min_fde = torch.empty(n, dtype=pred.dtype, device=dev)
    idx = torch.empty(n, dtype=torch.int32, device=dev)
    idx2 = torch.empty(n, dtype=torch.int32, device=dev)
    fn = lib().mtp_best_of_k_reranked_f32 if pred.dtype == torch.float32 else lib().mtp_best_of_k_reranked_f64
    with torch.cuda.device(dev):
        check(fn(ptr(pred), ptr(gt), ptr(order), order.shape[1], n, k, ktot, t, float(ade_div), int(mode), ptr(min_ade),
                 ptr(min_fde), ptr(idx), ptr(idx2), ptr(sums), stream_ptr()), "mtp_best_of_k_reranked")
    return dict(min_ade=min_ade, min_fde=min_fde, idx=idx, idx2=idx2, sums=sums)
