"""GPU tests at BASELINE.json's full sizes, through size-independent properties (the oracle
cannot finish these sizes in seconds): N = 10M x D = 128 x K = 1024 clustering steps and
1M agents x K = 100 x T = 12 best-of-K.  A slice of each result is still compared with the oracle.
"""
import numpy as np
import pytest
import torch

import kmeans_oracle as O

pytestmark = pytest.mark.gpu

N, D, K = 10_000_000, 128, 1024


@pytest.fixture(scope="module")
def big(gpu_device):
    g = torch.Generator(device=gpu_device)
    g.manual_seed(7)
    M = torch.randn(K, D, generator=g, device=gpu_device) * 0.5
    X = torch.empty((N, D), dtype=torch.float32, device=gpu_device)
    for s in range(0, N, 1 << 20):
        e = min(N, s + (1 << 20))
        z = torch.randint(0, K, (e - s,), generator=g, device=gpu_device)
        X[s:e] = torch.tanh(M[z] + 0.15 * torch.randn(e - s, D, generator=g, device=gpu_device))
    C = X[torch.randperm(1 << 20, generator=torch.Generator().manual_seed(1))[:K].to(gpu_device)].contiguous()
    yield X, C
    del X, C
    torch.cuda.empty_cache()


def test_assign_fullsize_properties(big, gpu_device):
    from pccs_accel import kernels as Kn
    X, C = big
    n_changed = torch.zeros(1, dtype=torch.float64, device=gpu_device)
    labels = Kn.assign(X, C, None, n_changed)
    assert int(n_changed.item()) == N                      # everything differs from the initial -1
    assert int(labels.min()) >= 0 and int(labels.max()) < K
    # idempotence: same centres again -> no label changes
    n_changed.zero_()
    again = Kn.assign(X, C, labels.clone(), n_changed)
    assert int(n_changed.item()) == 0 and torch.equal(again, labels)
    # a row permutation of the points permutes the labels (no dependence on position / tile)
    perm = torch.randperm(1 << 16, generator=torch.Generator().manual_seed(2)).to(gpu_device)
    sub = X[:1 << 16][perm].contiguous()
    assert torch.equal(Kn.assign(sub, C), labels[:1 << 16][perm])
    # the init rows are centres themselves: each is assigned to (a duplicate of) itself, distance 0
    own = Kn.assign(C.clone(), C)
    assert torch.equal(C[own.long()], C)
    # slices against the CPU oracle: head, an unaligned middle window, the ragged tail
    Ch = C.cpu().numpy()
    for lo, hi in ((0, 20000), (5_000_123, 5_020_123), (N - 12345, N)):
        want = O.assign(X[lo:hi].cpu().numpy(), Ch, backend="chain")
        assert np.array_equal(labels[lo:hi].cpu().numpy(), want)


def test_centroid_sums_fullsize_linearity(big, gpu_device):
    from pccs_accel import kernels as Kn
    X, C = big
    labels = Kn.assign(X, C)
    sums, counts = Kn.centroid_sums(X, labels, K)
    assert float(counts.sum().item()) == float(N)
    assert torch.equal(counts, torch.bincount(labels.long(), minlength=K).to(torch.float64))
    # sum over clusters of the per-cluster sums == column sums of X (double, order-independent)
    col = Kn.col_sums(X)
    assert torch.allclose(sums.sum(dim=0), col, rtol=1e-10, atol=1e-6)
    # splitting the points in two parts and adding the partial sums gives the same totals
    h = N // 2 + 77
    s2, c2 = Kn.centroid_sums(X[:h], labels[:h], K)
    Kn.centroid_sums(X[h:], labels[h:].contiguous(), K, s2, c2)
    assert torch.equal(c2, counts) and torch.allclose(s2, sums, rtol=1e-12, atol=1e-9)
    # one cluster against a float64 reference
    j = int(labels[12345].item())
    ref = X[labels == j].to(torch.float64).sum(dim=0)
    assert torch.allclose(sums[j], ref, rtol=1e-12, atol=1e-9)


def test_lloyd_fullsize_decreases_inertia(big, gpu_device):
    """Three Lloyd iterations at full size: inertia never increases, labels match the final centres."""
    from pccs_accel import kernels as Kn
    X, C = big
    mean = (Kn.col_sums(X) / N).to(torch.float32)
    Xc = X.clone()
    Kn.sub_rowvec_(Xc, mean)
    C0 = C.clone()
    Kn.sub_rowvec_(C0, mean)
    prev = None
    for iters in (1, 3):
        labels, inertia, centers, n_iter = Kn.lloyd(Xc, C0, max_iter=iters, tol_abs=0.0, check_every=1)
        assert n_iter == iters
        assert prev is None or inertia <= prev
        prev = inertia
    assert torch.equal(Kn.assign(Xc, centers), labels)
    assert np.isclose(Kn.inertia(Xc, centers, labels).item(), inertia, rtol=1e-12)
    del Xc


def test_seeding_fullsize_sharded_equals_fused(big, gpu_device):
    """k-means++ at the 8-GPU shard size (1.25M rows of the big matrix, 4 restarts, 8 local trials): the step-level
    sharded driver (candidate distances + count-below entries, here with one rank and no exchange) must pick the
    seeds of the fused one-GPU path, and the seeds must be valid distinct rows."""
    from pccs_accel import kernels as KK
    from pccs_accel import seeding
    X, _ = big
    n, k = 1_250_000, 17
    Xc = X[:n].clone()
    KK.sub_rowvec_(Xc, (KK.col_sums(Xc) / n).to(torch.float32))
    rs = np.random.RandomState(3)
    draws = [(c, np.ascontiguousarray(u[:k - 1])) for c, u in (seeding.draw(rs, n, K) for _ in range(4))]
    for order in ("f64", "openblas"):  # double potentials; float32 ones in the reference stack's order (block sums)
        c_f, i_f = seeding.kmeans_plusplus_batched(Xc, k, draws, pot_order=order)
        c_s, i_s = seeding.kmeans_plusplus_sharded(Xc, 0, n, k, draws, pot_order=order)
        assert torch.equal(i_f, i_s) and torch.equal(c_f, c_s)
    idx = i_f.cpu().numpy()
    assert idx.min() >= 0 and idx.max() < n and all(len(set(r)) == k for r in idx)
    assert torch.equal(c_f[2], Xc[i_f[2]])


def test_best_of_k_fullsize_properties(gpu_device):
    from pccs_accel import kernels as Kn
    n, k, t = 1_000_000, 100, 12
    g = torch.Generator(device=gpu_device)
    g.manual_seed(11)
    gt = torch.cumsum(torch.randn(n, t, 2, generator=g, device=gpu_device) * 0.3, dim=1).contiguous()
    pred = torch.empty((n, k, t, 2), dtype=torch.float32, device=gpu_device)
    for s in range(0, n, 1 << 16):
        e = min(n, s + (1 << 16))
        pred[s:e] = gt[s:e, None] + torch.randn(e - s, k, t, 2, generator=g, device=gpu_device) * 0.2
    pred[123, 57] = gt[123]                                # a perfect candidate: ADE = FDE = 0 at index 57
    r0 = Kn.best_of_k(pred, gt, mode=0)
    r1 = Kn.best_of_k(pred, gt, mode=1)
    r2 = Kn.best_of_k(pred, gt, mode=2)
    assert int(r0["idx"][123]) == 57 and float(r0["min_ade"][123]) == 0.0 and float(r1["min_fde"][123]) == 0.0
    # equal-focus takes both minima independently: it can only be better than either priority
    assert torch.equal(r2["min_ade"], r0["min_ade"]) and torch.equal(r2["min_fde"], r1["min_fde"])
    assert bool((r0["min_fde"] >= r2["min_fde"]).all()) and bool((r1["min_ade"] >= r2["min_ade"]).all())
    assert torch.equal(r2["idx"], r0["idx"]) and torch.equal(r2["idx2"], r1["idx"])
    # the float64 sums are the sums of the per-agent outputs
    assert np.isclose(r0["sums"][0].item(), r0["min_ade"].double().sum().item(), rtol=1e-12)
    assert np.isclose(r1["sums"][1].item(), r1["min_fde"].double().sum().item(), rtol=1e-12)
    # reversing the candidate order maps the argmin index (no exact ties in random data)
    sl = slice(0, 50_000)
    rr = Kn.best_of_k(pred[sl].flip(1).contiguous(), gt[sl].contiguous(), mode=0)
    assert torch.equal(rr["min_ade"], r0["min_ade"][sl])
    assert torch.equal(rr["idx"], (k - 1 - r0["idx"][sl]).to(torch.int32))
    # best_of < K equals scoring the truncated tensor
    rb = Kn.best_of_k(pred[sl].contiguous(), gt[sl].contiguous(), mode=1, best_of=20)
    rt = Kn.best_of_k(pred[sl, :20].contiguous(), gt[sl].contiguous(), mode=1)
    assert torch.equal(rb["idx"], rt["idx"]) and torch.equal(rb["min_fde"], rt["min_fde"])
    # slices against the oracle
    for lo, hi in ((0, 3000), (n - 1777, n)):
        w = O.best_of_k(pred[lo:hi].cpu().numpy(), gt[lo:hi].cpu().numpy(), mode=0)
        assert np.array_equal(r0["idx"][lo:hi].cpu().numpy(), w["idx"])
        assert np.allclose(r0["min_ade"][lo:hi].cpu().numpy(), w["min_ade"], rtol=1e-6)


def test_more_than_2_31_elements(gpu_device):
    """N * D beyond 2^31 elements (36M x 64 = 2.3e9 floats, 9.2 GB): every row offset must be 64-bit.  E-step slices
    at the head, across the 2^31-element line and at the ragged tail against the oracle; sums, column sums, inertia and
    the in-place centring through linearity between the whole array and its two halves."""
    from pccs_accel import kernels as Kn
    n, d, k = 36_000_003, 64, 300
    g = torch.Generator(device=gpu_device)
    g.manual_seed(11)
    M = torch.randn(k, d, generator=g, device=gpu_device) * 0.5
    X = torch.empty((n, d), dtype=torch.float32, device=gpu_device)
    for s in range(0, n, 1 << 21):
        e = min(n, s + (1 << 21))
        z = torch.randint(0, k, (e - s,), generator=g, device=gpu_device)
        X[s:e] = torch.tanh(M[z] + 0.15 * torch.randn(e - s, d, generator=g, device=gpu_device))
    C = X[torch.randperm(1 << 20, generator=torch.Generator().manual_seed(1))[:k].to(gpu_device)].contiguous()
    line = (1 << 31) // d  # the row whose first element is element 2^31
    labels = Kn.assign(X, C)
    assert int(labels.min()) >= 0 and int(labels.max()) < k
    Ch = C.cpu().numpy()
    for lo, hi in ((0, 10000), (line - 5000, line + 5000), (n - 7777, n)):
        want = O.assign(X[lo:hi].cpu().numpy(), Ch, backend="chain")
        assert np.array_equal(labels[lo:hi].cpu().numpy(), want), (lo, hi)
    # sums / counts: whole = head + tail, cut just below the line
    cut = line - 12345
    s_all, c_all = Kn.centroid_sums(X, labels, k)
    s_h, c_h = Kn.centroid_sums(X[:cut], labels[:cut].contiguous(), k)
    s_t, c_t = Kn.centroid_sums(X[cut:], labels[cut:].contiguous(), k)
    assert int(c_all.sum().item()) == n and torch.equal(c_all, c_h + c_t)
    assert torch.allclose(s_all, s_h + s_t, rtol=1e-12, atol=1e-9)
    ws, wc = O.accumulate_f64(X[n - 50000:].cpu().numpy(), labels[n - 50000:].cpu().numpy(), k)
    s_e, c_e = Kn.centroid_sums(X[n - 50000:], labels[n - 50000:].contiguous(), k)
    assert np.array_equal(c_e.cpu().numpy(), wc) and np.allclose(s_e.cpu().numpy(), ws, rtol=1e-12, atol=1e-12)
    # column sums and inertia the same way
    col = Kn.col_sums(X)
    assert torch.allclose(col, Kn.col_sums(X[:cut]) + Kn.col_sums(X[cut:]), rtol=1e-12, atol=1e-9)
    i_all = Kn.inertia(X, C, labels)
    i_parts = Kn.inertia(X[:cut], C, labels[:cut].contiguous()) + Kn.inertia(X[cut:], C, labels[cut:].contiguous())
    assert torch.allclose(torch.as_tensor(i_all, dtype=torch.float64), torch.as_tensor(i_parts, dtype=torch.float64), rtol=1e-12)
    # in-place centring: the rows beyond the line move by the mean too
    mean = (col / n).to(torch.float32)
    tail_before = X[n - 1000:].clone()
    mid_before = X[line - 500:line + 500].clone()
    Kn.sub_rowvec_(X, mean)
    assert torch.equal(X[n - 1000:], tail_before - mean) and torch.equal(X[line - 500:line + 500], mid_before - mean)
    del X
    torch.cuda.empty_cache()
