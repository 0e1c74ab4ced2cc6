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


def make_scene_trajectories(seed, n, t=20, extent=6, step=700, still_every=9, dtype=np.float32):
    """[n, t, 2] absolute positions of pedestrians sharing one scene (metres, exact dyadics):
    random start inside an ``extent`` x ``extent`` square, one of four common "flows" (mean step)
    per pedestrian plus jitter, so that many neighbours pass the speed / angle filter.
    Every ``still_every``-th pedestrian stands still (speed exactly 0) and every 13th repeats the
    pedestrian before it (identical speed and angle) -- the cases the neighbour predicate of
    PCCSNet/utils/data_process.py:148-158 treats specially."""
    rs = np.random.RandomState(seed)
    start = rs.randint(0, extent * 4096 + 1, size=(n, 1, 2))
    flows = rs.randint(-step, step + 1, size=(4, 1, 2))
    mean = flows[rs.randint(0, 4, size=n)] + rs.randint(-step // 16, step // 16 + 1, size=(n, 1, 2))
    jit = rs.randint(-step // 8, step // 8 + 1, size=(n, t, 2))
    steps = mean + jit
    steps[::still_every] = 0
    pos = start + np.cumsum(steps, axis=1)
    for i in range(13, n, 13):
        pos[i] = pos[i - 1] + rs.randint(-2000, 2001, size=(1, 2))  # same motion, shifted a little
    return np.ascontiguousarray((pos.astype(np.float64) / _Q).astype(dtype))


def make_displacements(seed, n, k, t, step=900, dtype=np.float32):
    """[n, k, t, 2] per-step displacements like the decoder's raw output (exact dyadics)."""
    rs = np.random.RandomState(seed)
    d = rs.randint(-step, step + 1, size=(n, k, t, 2))
    return np.ascontiguousarray((d.astype(np.float64) / _Q).astype(dtype))


def make_candidate_sets(seed, n, m, t, n_modes=9, dtype=np.float64):
    """Stored prediction sets for the re-ranking step: pred [n, m, t, 2] drawn around ``n_modes``
    futures per agent (so many candidates end within the NMS radius of each other), gt future
    [n, t, 2], verification scores [n, m] float32 with heavy ties after binning."""
    rs = np.random.RandomState(seed)
    modes = np.cumsum(rs.randint(-1500, 1501, size=(n, n_modes, t, 2)), axis=2)
    z = rs.randint(0, n_modes, size=(n, m))
    e = rs.randint(-500, 501, size=(n, m, t, 2))
    pred = modes[np.arange(n)[:, None], z] + e
    gt = modes[:, 0] + rs.randint(-300, 301, size=(n, t, 2))
    scores = rs.randint(-60 * 8, 20 * 8 + 1, size=(n, m)).astype(np.float64) / 8.0
    return (np.ascontiguousarray((pred.astype(np.float64) / _Q).astype(dtype)),
            np.ascontiguousarray((gt.astype(np.float64) / _Q).astype(dtype)),
            np.ascontiguousarray(scores.astype(np.float32)))
