"""The step-level seeding entry points (mtp_kpp_row_sqnorm_f64, mtp_kpp_candidate_dists_f32,
mtp_kpp_count_below_f32) through the C-ABI against NumPy / the CPU oracle's restatement of sklearn's upcast
_euclidean_distances (metrics/pairwise.py:391-442) and searchsorted(stable_cumsum(...)) (cluster/_kmeans.py:234-238)."""
import numpy as np
import pytest
import torch

import datagen
import kmeans_oracle as O

pytestmark = pytest.mark.gpu


def dev(a, device):
    return torch.from_numpy(np.ascontiguousarray(a)).to(device)


@pytest.mark.parametrize("n,d,b,l", [(1000, 96, 1, 7), (4133, 128, 3, 8), (257, 64, 6, 1), (5000, 96, 20, 7), (129, 16, 2, 3)])
def test_candidate_dists_match_oracle(n, d, b, l, gpu_device):
    from pccs_accel.seeding import DeviceSeedOps
    ops = DeviceSeedOps()
    X = datagen.make_encodings(77 + n, n, d, 9)
    X = X - X.mean(axis=0)
    rng = np.random.default_rng(n)
    cand = X[rng.integers(0, n, size=(b, l))]                       # [b, l, d] rows of X, like the real candidates
    closest = (rng.random((b, n)) * 3).astype(np.float32)
    Xd = dev(X, gpu_device)
    yy = ops.row_sqnorm(Xd)
    assert np.allclose(yy.cpu().numpy()[:n], (X.astype(np.float64) ** 2).sum(axis=1), rtol=1e-14)
    # ... and in the double summation order of NumPy's einsum (sklearn's row_norms on the upcast rows), bit for bit
    assert np.array_equal(yy.cpu().numpy()[:n].view(np.uint64), O.row_sqnorm_upcast_f64(X).view(np.uint64))
    for prev in (None, closest):
        rows, pot = ops.candidate_dists(Xd, yy, dev(cand, gpu_device), None if prev is None else dev(prev, gpu_device))
        rows, pot = rows.cpu().numpy(), pot.cpu().numpy()
        for r in range(b):
            want = O._euclidean_distances_sq_f32(cand[r], X)
            if prev is not None:
                np.minimum(prev[r][None, :], want, out=want)
            # float32 storage of a double result: identical unless the double lands within an ulp of a float32
            # rounding boundary (the oracle's dgemm order is NumPy's, the kernel's is the ascending fma chain)
            # (and the l distances of the candidates to themselves, which are cancellation noise around 0)
            bad = rows[r] != want
            assert bad.mean() <= 1.0 / n + 1e-4
            assert np.allclose(rows[r], want, rtol=3e-7, atol=1e-9)
            assert np.allclose(pot[r], rows[r].astype(np.float64).sum(axis=1), rtol=1e-12)


@pytest.mark.parametrize("n,b,l", [(1, 1, 1), (127, 2, 7), (128, 1, 8), (129, 3, 16), (100000, 20, 8), (131072 + 5, 2, 4)])
def test_count_below_is_searchsorted_of_cumsum(n, b, l, gpu_device):
    from pccs_accel.seeding import DeviceSeedOps
    ops = DeviceSeedOps()
    rng = np.random.default_rng(n + l)
    closest = (rng.random((b, n)) ** 3).astype(np.float32)
    closest[:, rng.integers(0, n, size=max(n // 10, 1))] = 0.0      # plateaus: equal cumulative sums
    cs = np.cumsum(closest.astype(np.float64), axis=1)
    thr = rng.random((b, l)) * cs[:, -1:] * 1.1 - 0.05 * cs[:, -1:]  # some below 0, some beyond the total
    thr[0, 0] = 0.0
    if l > 1:
        thr[-1, -1] = cs[-1, -1] * 2.0
    got = ops.count_below(dev(closest, gpu_device), dev(thr, gpu_device)).cpu().numpy()
    want = np.stack([np.searchsorted(cs[r], thr[r]) for r in range(b)])
    # the kernel's prefix is a block-sum tree, NumPy's a running sum: a threshold within ~1e-16 relative of a
    # cumulative sum may fall on the other side; none of these random thresholds does
    assert np.array_equal(got, want)


def test_seeding_steps_on_a_rank_without_rows(gpu_device):
    from pccs_accel.seeding import DeviceSeedOps
    ops = DeviceSeedOps()
    X = torch.empty((0, 96), dtype=torch.float32, device=gpu_device)
    yy = ops.row_sqnorm(X)
    cand = torch.randn(2, 7, 96, device=gpu_device)
    rows, pot = ops.candidate_dists(X, yy, cand, None)
    assert rows.shape == (2, 7, 0) and float(pot.abs().sum()) == 0.0
    cnt = ops.count_below(torch.empty((2, 0), dtype=torch.float32, device=gpu_device),
                          torch.ones((2, 7), dtype=torch.float64, device=gpu_device))
    assert int(cnt.abs().sum()) == 0


def test_candidate_dists_rejects_bad_shapes(gpu_device):
    from pccs_accel._lib import MtpError
    from pccs_accel.seeding import DeviceSeedOps
    ops = DeviceSeedOps()
    X = torch.randn(100, 40, device=gpu_device)                     # d not a multiple of 16
    with pytest.raises(MtpError):
        ops.candidate_dists(X, ops.row_sqnorm(X), torch.randn(1, 3, 40, device=gpu_device), None)
    X = torch.randn(100, 32, device=gpu_device)
    with pytest.raises(MtpError):
        ops.candidate_dists(X, ops.row_sqnorm(X), torch.randn(1, 9, 32, device=gpu_device), None)   # > 8 candidates
