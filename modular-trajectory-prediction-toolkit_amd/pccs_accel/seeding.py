"""k-means++ seeding on the device, replaying scikit-learn's draws.

Restates sklearn cluster/_kmeans.py:174-272 (``_kmeans_plusplus``, unit sample weights) as
KMeans.fit calls it for every one of the ``n_init`` restarts (PCCSNet/utils/utility.py:120-121
hard-codes n_init=20).  The random numbers come from the same NumPy ``RandomState`` in the same
order (one ``choice`` for the first centre, then ``uniform(size=2+int(ln K))`` per further
centre) and are uploaded once; everything that touches the N points runs in the HIP kernels of
csrc/seeding.hip (``mtp_kmeans_plusplus_f32``) without a host sync: candidate ids, potentials
and the closest-distance rows never leave the device.

Distances follow sklearn.metrics.pairwise._euclidean_distances for float32 input
(pairwise.py:391-442, :582-653): float64 arithmetic (-2 x.y + ||x||^2 + ||y||^2), float32
storage, clamped at 0.  Potentials are summed in float64 and rounded to float32 (sklearn's are
float32 BLAS sums, so a draw that lands within ~1e-6 relative of a cumulative-sum boundary can
pick the neighbouring point: whole-fit seeded parity is statistical at N ~ 1e5).
"""
import numpy as np
import torch

from ._lib import check, lib, ptr, stream_ptr


def kmeans_plusplus(Xc, n_clusters, random_state, n_local_trials=None):
    """Returns (centers float32 [K, D] device tensor, indices int64 [K] device tensor)."""
    if not (isinstance(Xc, torch.Tensor) and Xc.is_cuda and Xc.dtype == torch.float32 and Xc.dim() == 2
            and Xc.is_contiguous()):
        raise TypeError("kmeans_plusplus expects a contiguous float32 HIP device tensor [N, D]")
    n, d = Xc.shape
    dev = Xc.device
    if n_local_trials is None:
        n_local_trials = 2 + int(np.log(n_clusters))
    center_id, u = draw(random_state, n, n_clusters, n_local_trials)
    u_dev = torch.from_numpy(np.ascontiguousarray(u.reshape(-1))).to(dev) if u.size else None
    indices = torch.empty(n_clusters, dtype=torch.int64, device=dev)
    centers = torch.empty((n_clusters, d), dtype=torch.float32, device=dev)
    ws_bytes = int(lib().mtp_kpp_workspace_bytes(n, d, n_local_trials))
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        check(lib().mtp_kmeans_plusplus_f32(ptr(Xc), n, d, n_clusters, n_local_trials, center_id, ptr(u_dev),
                                            ptr(indices), ptr(centers), ptr(ws), ws_bytes, stream_ptr()),
              "mtp_kmeans_plusplus_f32")
    return centers, indices


_CDF = {}


def _uniform_cdf(n):
    """The cdf ``RandomState.choice(n, p=w / w.sum())`` builds for unit float32 weights
    (numpy/random/mtrand.pyx: p -> float64, cdf = p.cumsum(); cdf /= cdf[-1]); it only depends on n."""
    if n not in _CDF:
        _CDF.clear()
        w = np.ones(n, dtype=np.float32)
        cdf = (w / w.sum()).astype(np.float64).cumsum()
        cdf /= cdf[-1]
        _CDF[n] = cdf
    return _CDF[n]


def draw(random_state, n, n_clusters, n_local_trials=None):
    """The random numbers one ``_kmeans_plusplus`` call consumes, in sklearn's order:
    (first centre index, uniform [K-1, L]).  ``choice(n, p=...)`` is one ``random_sample()``
    looked up in the cdf (side='right'); done that way the O(n) cdf is built once, not per restart."""
    if n_local_trials is None:
        n_local_trials = 2 + int(np.log(n_clusters))
    center_id = int(_uniform_cdf(n).searchsorted(random_state.random_sample(), side="right"))
    u = random_state.uniform(size=(max(n_clusters - 1, 0), n_local_trials))
    return center_id, u


def kmeans_plusplus_batched(Xc, n_clusters, draws):
    """Seeding of len(draws) restarts in shared launches.  ``draws``: list of ``draw()`` results,
    one per restart, taken from the RandomState in restart order.  Returns (centers float32
    [B, K, D], indices int64 [B, K]) device tensors."""
    if not (isinstance(Xc, torch.Tensor) and Xc.is_cuda and Xc.dtype == torch.float32 and Xc.dim() == 2
            and Xc.is_contiguous()):
        raise TypeError("kmeans_plusplus_batched expects a contiguous float32 HIP device tensor [N, D]")
    n, d = Xc.shape
    dev = Xc.device
    b = len(draws)
    l = draws[0][1].shape[1] if n_clusters > 1 else 2 + int(np.log(n_clusters))
    ids = np.asarray([c for c, _ in draws], dtype=np.int64)
    if ids.min() < 0 or ids.max() >= n:
        raise ValueError("first-centre index out of range")
    ids_dev = torch.from_numpy(ids).to(dev)
    u_dev = None
    if n_clusters > 1:
        u = np.ascontiguousarray(np.stack([u for _, u in draws]).reshape(-1))
        u_dev = torch.from_numpy(u).to(dev)
    indices = torch.empty((b, n_clusters), dtype=torch.int64, device=dev)
    centers = torch.empty((b, n_clusters, d), dtype=torch.float32, device=dev)
    ws_bytes = int(lib().mtp_kpp_batched_workspace_bytes(n, d, l, b))
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        check(lib().mtp_kmeans_plusplus_batched_f32(ptr(Xc), n, d, n_clusters, l, b, ptr(ids_dev), ptr(u_dev), ptr(indices),
                                                    ptr(centers), ptr(ws), ws_bytes, stream_ptr()),
              "mtp_kmeans_plusplus_batched_f32")
    return centers, indices
