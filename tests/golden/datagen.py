"""Platform-independent synthetic inputs for the golden fixtures and the parity tests.

Everything is built from ``numpy.random.RandomState(seed).randint`` (raw Mersenne-Twister
integers) and exact dyadic scaling, so the same arrays come out bit-for-bit on the build
container and on the GPU box -- no libm call, no SIMD-dispatch-dependent ufunc.  The
fixtures therefore store only seeds/shapes and the expected outputs, not the inputs.

The shapes mimic the reference's data: "encodings" are a K-modal mixture in (-1, 1) like
bi-LSTM outputs (PCCSNet/models/components.py:133-150), trajectories are cumulative sums
of small steps like ``transform_trajectory`` output (PCCSNet/utils/data_process.py:111-122).
"""
import numpy as np

_Q = 4096.0  # all values are integers / 4096 -> exact in float32 and float64


def make_encodings(seed, n, d, k_true, spread=2600, noise=160, dtype=np.float32):
    """[n, d] mixture of ``k_true`` modes: (M[z] + e) / 4096 with integer M, e."""
    rs = np.random.RandomState(seed)
    modes = rs.randint(-spread, spread + 1, size=(k_true, d))
    z = rs.randint(0, k_true, size=n)
    e = rs.randint(-noise, noise + 1, size=(4, n, d)).sum(axis=0)  # ~ bell shaped
    x = (modes[z] + e).astype(np.float64) / _Q
    return np.ascontiguousarray(x.astype(dtype))


def make_init_indices(seed, n, k):
    """k distinct row indices (Fisher-Yates on randint only)."""
    rs = np.random.RandomState(seed)
    perm = np.arange(n)
    for i in range(k):
        j = i + int(rs.randint(0, n - i))
        perm[i], perm[j] = perm[j], perm[i]
    return perm[:k].copy()


def make_trajectories(seed, n, k, t, step=1200, noise=800, dtype=np.float32, scale=1.0):
    """gt [n, t, 2] = cumsum of steps; pred [n, k, t, 2] = gt + noise.  All exact dyadics
    (``scale`` must be a power of two to keep that property)."""
    rs = np.random.RandomState(seed)
    steps = rs.randint(-step, step + 1, size=(n, t, 2))
    gt = np.cumsum(steps, axis=1)
    e = rs.randint(-noise, noise + 1, size=(n, k, t, 2))
    pred = gt[:, None] + e
    gt = (gt.astype(np.float64) / _Q * scale).astype(dtype)
    pred = (pred.astype(np.float64) / _Q * scale).astype(dtype)
    return np.ascontiguousarray(pred), np.ascontiguousarray(gt)
