"""k-means++ seeding on the device, replaying scikit-learn's draws.

Restates sklearn cluster/_kmeans.py:174-272 (``_kmeans_plusplus``, unit sample weights) as
KMeans.fit calls it for every one of the ``n_init`` restarts (PCCSNet/utils/utility.py:120-121
hard-codes n_init=20).  The random numbers come from the same NumPy ``RandomState`` in the same
order (one ``choice`` for the first centre, then ``uniform(size=2+int(ln K))`` per further
centre), everything that touches the N points runs on the GPU, and no step needs the host:
the candidate ids, potentials and closest-distance vector never leave the device.

Distances follow sklearn.metrics.pairwise._euclidean_distances for float32 input
(pairwise.py:391-442, :582-653): float64 arithmetic (-2 x.y + ||x||^2 + ||y||^2), float32
storage, clamped at 0.  Potentials are summed in float64 and rounded to float32 (sklearn's are
float32 BLAS sums, so a draw that lands within ~1e-7 relative of a cumulative-sum boundary can
pick the neighbouring point: whole-fit seeded parity is statistical).
"""
import numpy as np
import torch


def _dist_sq_f32(Y64, yy, X64, xx):
    d = torch.addmm(xx.unsqueeze(0) + yy.unsqueeze(1), Y64, X64.t(), beta=1.0, alpha=-2.0)
    return d.to(torch.float32).clamp_(min=0)


def kmeans_plusplus(Xc, n_clusters, random_state, n_local_trials=None):
    """Returns (centers float32 [K, D] device tensor, indices int64 [K] device tensor)."""
    n, d = Xc.shape
    dev = Xc.device
    if n_local_trials is None:
        n_local_trials = 2 + int(np.log(n_clusters))
    w = np.ones(n, dtype=np.float32)
    center_id = int(random_state.choice(n, p=w / w.sum()))
    u = random_state.uniform(size=(max(n_clusters - 1, 0), n_local_trials))
    u = torch.from_numpy(np.ascontiguousarray(u)).to(dev)

    X64 = Xc.to(torch.float64)
    xx = (X64 * X64).sum(dim=1)
    indices = torch.full((n_clusters,), -1, dtype=torch.int64, device=dev)
    indices[0] = center_id
    closest = _dist_sq_f32(X64[center_id:center_id + 1], xx[center_id:center_id + 1], X64, xx)[0]
    pot = closest.sum(dtype=torch.float64).to(torch.float32)
    for c in range(1, n_clusters):
        rand_vals = u[c - 1] * pot.to(torch.float64)
        cum = torch.cumsum(closest.to(torch.float64), dim=0)
        cand = torch.searchsorted(cum, rand_vals).clamp_(max=n - 1)
        dist = _dist_sq_f32(X64[cand], xx[cand], X64, xx)
        torch.minimum(closest.unsqueeze(0), dist, out=dist)
        pots = dist.sum(dim=1, dtype=torch.float64).to(torch.float32)
        b = torch.argmin(pots)
        pot = pots[b]
        closest = dist[b]
        indices[c] = cand[b]
    return Xc[indices].contiguous(), indices
