"""GPU parity tests: the HIP path (through the C-ABI) against the CPU oracle and the golden
vectors generated from the reference.  Everything here is marked ``gpu``.

Bars (north_star): integer outputs (labels, argmin indices, counts, n_iter) bit-exact --
label mismatches are tolerated only where the float64-recomputed distance gap is < 1e-6
relative, and are counted; centroids / inertia / ADE / FDE within 1e-5 relative.
"""
import numpy as np
import pytest
import torch

import datagen
import kmeans_oracle as O
from conftest import G1_NAMES, g1_inputs, load_golden

pytestmark = pytest.mark.gpu

RTOL = 1e-5


def dev(a, gpu_device):
    return torch.from_numpy(np.ascontiguousarray(a)).to(gpu_device)


def assert_labels_match(X, C, got, want, allow_near_ties=True):
    got = np.asarray(got)
    want = np.asarray(want)
    r = O.classify_label_mismatches(X, C, got, want)
    if not allow_near_ties:
        assert r["n_mismatch"] == 0, r
    assert r["n_bad"] == 0, r
    return r["n_mismatch"]


# ------------------------------------------------------------------------------------------
# E-step
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", G1_NAMES)
def test_assign_bit_exact_vs_oracle_and_sklearn(name, gpu_device):
    """MFMA E-step == oracle chain E-step == sklearn's first-iteration labels, bit for bit."""
    from pccs_accel import kernels as K
    g = load_golden(name)
    X, C0 = g1_inputs(g)
    xm = g["x_mean"]
    Xc, Cc = X - xm, C0 - xm
    labels = K.assign(dev(Xc, gpu_device), dev(Cc, gpu_device)).cpu().numpy()
    assert labels.dtype == np.int32
    want = O.assign(Xc, Cc, backend="chain")
    assert np.array_equal(labels, want)
    assert np.array_equal(labels, g["iter1_labels"])


@pytest.mark.parametrize("n,d,k", [(1, 96, 1), (63, 96, 5), (257, 128, 33), (1000, 64, 70), (300, 8, 7),
                                   (129, 2, 3), (500, 100, 40), (77, 7, 77)])
def test_assign_ragged_shapes(n, d, k, gpu_device):
    """Tail rows, K not a multiple of the 32-centroid tile, generic-kernel dims; n_changed."""
    from pccs_accel import kernels as K
    X = datagen.make_encodings(100 + n, n, d, max(1, k // 2), 1500, 700)
    C = X[datagen.make_init_indices(7, n, k)].copy()
    prev = torch.full((n,), -1, dtype=torch.int32, device=gpu_device)
    n_changed = torch.zeros(1, dtype=torch.float64, device=gpu_device)
    labels = K.assign(dev(X, gpu_device), dev(C, gpu_device), prev, n_changed).cpu().numpy()
    want = O.assign(X, C, backend="chain")
    assert np.array_equal(labels, want)
    assert int(n_changed.item()) == n
    # second call with the same centres: nothing changes
    n_changed.zero_()
    K.assign(dev(X, gpu_device), dev(C, gpu_device), prev, n_changed)
    assert int(n_changed.item()) == 0


@pytest.mark.parametrize("delta", [-33, 0, 1, 257, 2199 * 32 - 7, 3000 * 32])
def test_assign_split_launch_boundaries(delta, gpu_device):
    """A single E-step is issued as two launches (whole rounds of 2 x CUs workgroups of 256 points, then the
    remaining points with workgroups of 4 / 8 / 12 / 16 waves): point counts just below, at and above whole rounds,
    and remainders that select every tail workgroup size, against the oracle; n_changed over both launches."""
    from pccs_accel import kernels as K
    per_round = 2 * torch.cuda.get_device_properties(gpu_device).multi_processor_count * 256
    n, d, k = per_round + delta, 64, 40
    X = datagen.make_encodings(500 + (delta % 97), n, d, 20, 1500, 700)
    C = X[datagen.make_init_indices(9, n, k)].copy()
    prev = torch.full((n,), -1, dtype=torch.int32, device=gpu_device)
    n_changed = torch.zeros(1, dtype=torch.float64, device=gpu_device)
    labels = K.assign(dev(X, gpu_device), dev(C, gpu_device), prev, n_changed).cpu().numpy()
    assert np.array_equal(labels, O.assign(X, C, backend="chain"))
    assert int(n_changed.item()) == n


def test_assign_exact_ties_take_lowest_index(gpu_device):
    """Duplicate centroids: strict '<' keeps the first (sklearn _k_means_lloyd.pyx:205-213)."""
    from pccs_accel import kernels as K
    X = datagen.make_encodings(5, 512, 96, 6)
    base = X[datagen.make_init_indices(9, 512, 40)].copy()
    C = np.concatenate([base, base[::-1]], axis=0)  # every centre appears twice: j and 79-j
    labels = K.assign(dev(X, gpu_device), dev(C, gpu_device)).cpu().numpy()
    assert labels.max() < 40
    assert np.array_equal(labels, O.assign(X, C, backend="chain"))


def test_row_sqnorm_einsum_bits(gpu_device):
    from pccs_accel import kernels as K
    rs = np.random.RandomState(0)
    for d in (1, 2, 5, 16, 17, 96, 100, 128):
        A = (rs.randint(-4000, 4000, size=(300, d)) / 4096.0).astype(np.float32)
        out = K.row_sqnorm(dev(A, gpu_device)).cpu().numpy()
        assert np.array_equal(out, np.einsum("ij,ij->i", A, A)), d


# ------------------------------------------------------------------------------------------
# M-step pieces
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", G1_NAMES)
def test_centroid_sums_and_get_modality(name, gpu_device):
    from pccs_accel import kernels as K
    from pccs_accel.utils.utility import get_modality
    g = load_golden(name)
    X, _ = g1_inputs(g)
    k = int(g["K"])
    labels = g["labels"]
    sums, counts = K.centroid_sums(dev(X, gpu_device), dev(labels, gpu_device), k)
    ws, wc = O.accumulate_f64(X, labels, k)
    assert np.array_equal(counts.cpu().numpy(), wc)
    assert np.allclose(sums.cpu().numpy(), ws, rtol=1e-12, atol=1e-12)
    # drop-in get_modality: CPU tensor in (train_steps.py:356), float32 [K, D] out
    mod = get_modality(torch.from_numpy(X), labels, k)
    assert mod.dtype == torch.float32 and mod.device.type == "cpu"
    assert np.allclose(mod.numpy(), g["modality"], rtol=RTOL, atol=1e-7)
    assert np.allclose(mod.numpy(), O.get_modality(X, labels, k), rtol=1e-6, atol=1e-7)


def test_get_modality_empty_cluster_keyerror(gpu_device):
    from pccs_accel.utils.utility import get_modality
    X = datagen.make_encodings(1, 64, 8, 2)
    with pytest.raises(KeyError):
        get_modality(torch.from_numpy(X), np.zeros(64, dtype=np.int32), 2)


@pytest.mark.parametrize("n,d,k", [(1, 4, 1), (100, 12, 3), (5000, 96, 200), (3000, 128, 300), (999, 7, 5),
                                   (4000, 96, 450), (2500, 64, 700), (3000, 32, 1300)])
def test_centroid_sums_shapes(n, d, k, gpu_device):
    """Ranges (K*D*8 > LDS: the range kernel with its non-temporal row loads at d = 64 / 96 / 128 and with d as an
    argument), tiny inputs, d % 4 != 0 fallback."""
    from pccs_accel import kernels as K
    X = datagen.make_encodings(n + d, n, d, 3)
    labels = np.random.RandomState(n).randint(0, k, size=n).astype(np.int32)
    sums, counts = K.centroid_sums(dev(X, gpu_device), dev(labels, gpu_device), k)
    ws, wc = O.accumulate_f64(X, labels, k)
    assert np.array_equal(counts.cpu().numpy(), wc)
    assert np.allclose(sums.cpu().numpy(), ws, rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize("name", G1_NAMES[:2])
def test_inertia(name, gpu_device):
    from pccs_accel import kernels as K
    g = load_golden(name)
    X, _ = g1_inputs(g)
    v, pp = K.inertia(dev(X, gpu_device), dev(g["centers"], gpu_device), dev(g["labels"], gpu_device), per_point=True)
    want, wpp = O.inertia(X, g["centers"], g["labels"], return_per_point=True)
    assert np.isclose(v.item(), want, rtol=1e-6)
    assert np.allclose(pp.cpu().numpy(), wpp, rtol=1e-5, atol=1e-9)
    assert np.isclose(v.item(), float(g["inertia"]), rtol=RTOL)


# ------------------------------------------------------------------------------------------
# whole Lloyd runs
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", G1_NAMES)
@pytest.mark.parametrize("check_every", [1, 8])
def test_lloyd_fixed_init_vs_golden(name, check_every, gpu_device):
    """_kmeans_single_lloyd on the centred data: n_iter, labels, centres, inertia."""
    from pccs_accel import kernels as K
    g = load_golden(name)
    X, C0 = g1_inputs(g)
    xm = g["x_mean"]
    Xc = X - xm
    labels, inertia, centers, n_iter = K.lloyd(dev(Xc, gpu_device), dev(C0 - xm, gpu_device), max_iter=300,
                                               tol_abs=float(g["tol_abs"]), check_every=check_every)
    assert n_iter == int(g["n_iter"])
    centers = centers.cpu().numpy()
    n_mis = assert_labels_match(Xc, centers, labels.cpu().numpy(), g["labels"])
    assert n_mis == 0  # observed: bit-exact on every fixture
    assert np.allclose(centers + xm, g["centers"], rtol=RTOL, atol=1e-6)
    assert np.isclose(inertia, float(g["inertia"]), rtol=RTOL)
    # and against the oracle run end to end
    ol, oi, oc, on = O.kmeans_single_lloyd(Xc, C0 - xm, 300, float(g["tol_abs"]))
    assert on == n_iter and np.array_equal(ol, labels.cpu().numpy())


@pytest.mark.parametrize("name", G1_NAMES)
def test_kmeans_fit_fixed_init(name, gpu_device):
    """KMeans(init=C0, n_init=1).fit including the device-side mean / tolerance."""
    from pccs_accel.clustering import kmeans_fit
    g = load_golden(name)
    X, C0 = g1_inputs(g)
    fit = kmeans_fit(dev(X, gpu_device), int(g["K"]), init=C0)
    assert np.isclose(fit["tol_abs"], float(g["tol_abs"]), rtol=1e-5)
    assert fit["n_iter"] == int(g["n_iter"])
    centers = fit["centers"].cpu().numpy()
    assert_labels_match(X, centers, fit["labels"].cpu().numpy(), g["labels"])
    assert np.allclose(centers, g["centers"], rtol=RTOL, atol=1e-6)
    assert np.isclose(fit["inertia"], float(g["inertia"]), rtol=RTOL)


def test_max_iter_and_final_estep(gpu_device):
    """max_iter hit before convergence: one more E-step makes labels match the final centres."""
    from pccs_accel import kernels as K
    g = load_golden("g1_hard_n5000_d96_k64")
    X, C0 = g1_inputs(g)
    xm = g["x_mean"]
    Xc = X - xm
    labels, inertia, centers, n_iter = K.lloyd(dev(Xc, gpu_device), dev(C0 - xm, gpu_device), max_iter=3,
                                               tol_abs=float(g["tol_abs"]))
    ol, oi, oc, on = O.kmeans_single_lloyd(Xc, C0 - xm, 3, float(g["tol_abs"]))
    assert n_iter == on == 3
    assert np.array_equal(labels.cpu().numpy(), ol)
    assert np.allclose(centers.cpu().numpy(), oc, rtol=RTOL, atol=1e-6)
    assert np.array_equal(labels.cpu().numpy(), O.assign(Xc, centers.cpu().numpy()))


def test_kat_hand_computed_results(gpu_device):
    """sklearn/cluster/tests/test_k_means.py:62-80 through the generic (d=2) kernels."""
    from pccs_accel.clustering import kmeans_fit
    g = load_golden("g5_kat_results")
    fit = kmeans_fit(dev(g["X"], gpu_device), 2, init=g["init"])
    assert list(fit["labels"].cpu().numpy()) == [0, 0, 0, 0, 1, 1, 1, 1]
    assert fit["n_iter"] == 2
    assert np.allclose(fit["centers"].cpu().numpy(), [[0.125, 0], [0.875, 1]])
    assert np.isclose(fit["inertia"], 0.375)


def test_kat_relocated_clusters(gpu_device):
    """sklearn/cluster/tests/test_k_means.py:85-111: an empty cluster is relocated onto the
    farthest point (all four are equally far here; both outcomes are accepted by the KAT)."""
    from pccs_accel.clustering import kmeans_fit
    g = load_golden("g4_relocated_clusters")
    fit = kmeans_fit(dev(g["X"].astype(np.float32), gpu_device), 2, init=g["init"].astype(np.float32))
    labels = list(fit["labels"].cpu().numpy())
    centers = fit["centers"].cpu().numpy()
    assert fit["n_iter"] == 3 and np.isclose(fit["inertia"], 0.25)
    ok_a = labels == [0, 0, 1, 1] and np.allclose(centers, [[0.25, 0], [0.75, 1]])
    ok_b = labels == [1, 1, 0, 0] and np.allclose(centers, [[0.75, 1], [0.25, 0]])
    assert ok_a or ok_b


def test_relocation_on_mfma_path(gpu_device):
    """A centre placed far away starts empty (d=96, MFMA path): it must be relocated onto the
    point farthest from its centre, exactly as the oracle (== sklearn) does."""
    from pccs_accel import kernels as K
    X = datagen.make_encodings(77, 3000, 96, 10, 2000, 300)
    xm = X.mean(axis=0)
    Xc = X - xm
    C0 = Xc[datagen.make_init_indices(78, 3000, 12)].copy()
    C0[5] = 50.0  # nobody is closest to this one
    tol = float(O.tolerance(X))
    labels, inertia, centers, n_iter = K.lloyd(dev(Xc, gpu_device), dev(C0, gpu_device), tol_abs=tol)
    ol, oi, oc, on = O.kmeans_single_lloyd(Xc, C0, 300, tol)
    assert n_iter == on
    assert_labels_match(Xc, oc, labels.cpu().numpy(), ol, allow_near_ties=False)
    assert np.allclose(centers.cpu().numpy(), oc, rtol=RTOL, atol=1e-6)
    assert len(np.unique(labels.cpu().numpy())) == 12


def test_relocation_of_more_than_64_empty_clusters(gpu_device):
    """100 of 160 centres start unreachable: sklearn relocates every one of them in the first iteration
    (_k_means_common.pyx:167-211).  Single run and batched restarts against the oracle -- n_iter, labels,
    centres -- with the far points handed out by descending distance (sklearn's argpartition order is
    unspecified), and against sklearn's own order up to a renaming of the relocated clusters."""
    from pccs_accel import kernels as K
    n, d, k = 6001, 96, 160
    X = datagen.make_encodings(91, n, d, 30, 2000, 300)
    Xc = np.ascontiguousarray(X - X.mean(axis=0))
    C0 = Xc[datagen.make_init_indices(92, n, k)].copy()
    for j in range(100):
        C0[(3 * j) % k] = 30.0 + j
    tol = float(O.tolerance(X))
    assert len(np.unique(O.assign(Xc, C0))) <= 60                      # 100 clusters really start empty
    # exact against the product's documented variant (double sums, far points by descending distance): with 100
    # rows taken out of float sums the float-sum variant rounds differently enough to re-label a few points of
    # this far-from-converged start (9 of 6001 here, inertia 2.7e-5 apart) -- it is compared on n_iter / inertia
    ol, oi, oc, on = O.kmeans_single_lloyd(Xc, C0, 300, tol, far_order="descending", sums="f64")
    labels, inertia, centers, n_iter = K.lloyd(dev(Xc, gpu_device), dev(C0, gpu_device), tol_abs=tol)
    assert n_iter == on
    assert_labels_match(Xc, oc, labels.cpu().numpy(), ol, allow_near_ties=False)
    assert np.allclose(centers.cpu().numpy(), oc, rtol=RTOL, atol=1e-6)
    assert np.isclose(inertia, oi, rtol=RTOL)
    assert len(np.unique(labels.cpu().numpy())) == k
    # sklearn's own hand-out order: the same clustering with some of the relocated ids swapped
    sl, si, _, sn = O.kmeans_single_lloyd(Xc, C0, 300, tol, far_order="argpartition", sums="f64")
    assert sn == on and np.isclose(si, oi, rtol=1e-6)
    assert O.is_same_clustering(sl, ol, k) and O.is_same_clustering(ol, sl, k)
    fl, fi, _, fn = O.kmeans_single_lloyd(Xc, C0, 300, tol, far_order="argpartition")  # sklearn as it is, one thread
    assert fn == on and np.isclose(fi, oi, rtol=1e-3) and int((fl != sl).sum()) <= n // 100
    # the batched driver: two such restarts beside an ordinary one
    C0s = np.stack([C0, Xc[datagen.make_init_indices(93, n, k)], C0[::-1].copy()])
    lb, ib, cb, nb = K.lloyd_batched(dev(Xc, gpu_device), dev(C0s, gpu_device), tol_abs=tol)
    assert nb[0] == n_iter and torch.equal(lb[0], labels) and torch.equal(cb[0], centers)
    for r in (1, 2):
        l1, i1, c1, n1 = K.lloyd(dev(Xc, gpu_device), dev(C0s[r], gpu_device), tol_abs=tol)
        assert nb[r] == n1 and torch.equal(lb[r], l1) and torch.equal(cb[r], c1)



def test_relocation_serves_only_the_originally_empty_clusters(gpu_device):
    """sklearn fixes `empty_clusters = where(weight_in_clusters == 0)` before its loop
    (_k_means_common.pyx:176): when a far point is the only member of its cluster, that cluster drops to
    zero during the loop but is NOT served; the last originally empty cluster is."""
    from pccs_accel._lib import check, lib, ptr, stream_ptr
    k, d, m = 4, 2, 2
    sums = np.array([[0, 0], [7, 8], [0, 0], [10, 20]], dtype=np.float64)
    counts = np.array([0, 1, 0, 5], dtype=np.float64)          # clusters 0 and 2 are empty, cluster 1 is a singleton
    red = dev(np.concatenate([sums.ravel(), counts, [0, 0]]), gpu_device)
    far_val = dev(np.array([9.0, 4.0], dtype=np.float32), gpu_device)
    far_lab = dev(np.array([1, 3], dtype=np.int32), gpu_device)   # the farthest point IS cluster 1's only member
    far_rows = dev(np.array([[7, 8], [1, 2]], dtype=np.float32), gpu_device)
    check(lib().mtp_apply_relocation_f32(ptr(red), k, d, m, ptr(far_val), ptr(far_lab), ptr(far_rows), stream_ptr()),
          "mtp_apply_relocation_f32")
    out = red.cpu().numpy()
    assert out[k * d:k * d + k].tolist() == [1, 0, 1, 4]
    assert not np.signbit(out[k * d:k * d + k]).any()
    assert out[:k * d].reshape(k, d).tolist() == [[7, 8], [0, 0], [1, 2], [9, 18]]
    # fewer far points than empty clusters: the unserved mark is cleared again
    red = dev(np.concatenate([sums.ravel(), counts, [0, 0]]), gpu_device)
    far_lab = dev(np.array([3, -1], dtype=np.int32), gpu_device)
    check(lib().mtp_apply_relocation_f32(ptr(red), k, d, m, ptr(far_val), ptr(far_lab), ptr(far_rows), stream_ptr()),
          "mtp_apply_relocation_f32")
    out = red.cpu().numpy()
    assert out[k * d:k * d + k].tolist() == [1, 1, 0, 4] and not np.signbit(out[k * d:k * d + k]).any()


@pytest.mark.parametrize("d", [12, 64, 128, 300])
def test_far_points_rank_on_numpys_row_sums(d, gpu_device):
    """sklearn picks its far points by `distances = ((X - centers_old[labels]) ** 2).sum(axis=1)`
    (_k_means_common.pyx:183-187): float32 differences and squares, NumPy's pairwise sum along the row.  The search
    ranks on exactly those bits (descending, ties to the lowest index), so two points whose distances differ in the
    last bit rank as they do in sklearn."""
    from pccs_accel._lib import check, lib, ptr, stream_ptr
    n, k, m = 3000, 17, 40
    X = datagen.make_encodings(300 + d, n, d, 9, 1500, 700)
    X[1500] = X[7]        # exact ties between rows far apart
    X[2999] = X[8]
    C = X[datagen.make_init_indices(3, n, k)].copy()
    labels = O.assign(X, C, backend="chain")
    want_d = ((X - C[labels]) ** 2).sum(axis=1)
    order = np.lexsort((np.arange(n), -want_d))[:m]
    Xd, Cd, ld = dev(X, gpu_device), dev(C, gpu_device), dev(labels, gpu_device)
    val = torch.empty(m, dtype=torch.float32, device=gpu_device)
    idx = torch.empty(m, dtype=torch.int64, device=gpu_device)
    lab = torch.empty(m, dtype=torch.int32, device=gpu_device)
    rows = torch.empty((m, d), dtype=torch.float32, device=gpu_device)
    scratch = torch.empty(int(lib().mtp_far_points_scratch_bytes(n)), dtype=torch.uint8, device=gpu_device)
    check(lib().mtp_far_points_f32(ptr(Xd), n, d, ptr(Cd), ptr(ld), m, 0, ptr(val), ptr(idx), ptr(lab), ptr(rows),
                                   ptr(scratch), stream_ptr()), "mtp_far_points_f32")
    assert np.array_equal(idx.cpu().numpy(), order)
    assert np.array_equal(val.cpu().numpy(), want_d[order])          # the distances themselves, bit for bit
    assert np.array_equal(lab.cpu().numpy(), labels[order]) and np.array_equal(rows.cpu().numpy(), X[order])


def test_relocation_with_a_near_tie_at_the_last_far_point(gpu_device):
    """Found by tests/fuzz_gpu_parity.py (seed 6, round 31): 23 of 39 clusters start empty and the 23rd and 24th
    farthest points are 2e-6 apart -- closer than the difference between two float summation orders of the same
    distance.  With the ranking on NumPy's own row sums the run equals the oracle's again."""
    from pccs_accel import kernels as K
    rs = np.random.RandomState(1143311898)
    n, d, k = int(rs.randint(60, 2500)), int(rs.choice([64, 96, 128, 12])), int(rs.randint(3, 40))
    X = datagen.make_encodings(int(rs.randint(1 << 30)), n, d, max(1, k // 2), 1500, 700)
    X = X - X.mean(0)
    C0 = X[rs.choice(n, size=k, replace=False)].copy()
    n_far = int(rs.randint(1, k))
    C0[rs.choice(k, size=n_far, replace=False)] += np.float32(50.0)
    assert (n, d, k, n_far) == (535, 128, 39, 23)
    tol = float(O.tolerance(X))
    labels, inertia, centers, n_iter = K.lloyd(dev(X, gpu_device), dev(C0, gpu_device), max_iter=40, tol_abs=tol, check_every=4)
    ol, oi, oc, on = O.kmeans_single_lloyd(X, C0, 40, tol, sums="f64", far_order="descending")
    assert n_iter == on and np.array_equal(labels.cpu().numpy(), ol)
    assert np.allclose(centers.cpu().numpy(), oc, rtol=RTOL, atol=1e-6) and np.isclose(inertia, oi, rtol=RTOL)


# ------------------------------------------------------------------------------------------
# ADE / FDE and best-of-K
# ------------------------------------------------------------------------------------------
def _g6_inputs(g):
    pred, gt = datagen.make_trajectories(int(g["seed"]), int(g["n"]), int(g["k"]), int(g["t"]))
    pred[5, 3] = gt[5] + np.float32(1.0 / 4096)
    pred[5, 7] = pred[5, 3]
    pred[9, 2, -1] = gt[9, -1]
    pred[9, 11, -1] = gt[9, -1]
    return pred, gt


def test_metrics_dropin(gpu_device):
    """displacement_error / final_displacement_error with the reference's calling patterns."""
    from pccs_accel.utils.metrics import displacement_error, final_displacement_error
    g = load_golden("g6_ade_fde")
    pred, gt = _g6_inputs(g)
    tp, tg = dev(pred, gpu_device), dev(gt, gpu_device)
    t = int(g["t"])
    for i in (0, 5, 9, 63):
        A = displacement_error(tp[i], tg[i:i + 1]) / t          # train_steps.py:219
        F = final_displacement_error(tp[i][:, -1], tg[i:i + 1][:, -1])  # train_steps.py:220
        assert A.shape == (int(g["k"]),) and A.dtype == torch.float32 and A.is_cuda
        assert np.allclose(A.cpu().numpy(), g["ade"][i], rtol=RTOL)
        assert np.allclose(F.cpu().numpy(), g["fde"][i], rtol=RTOL)
    assert np.isclose(displacement_error(tp[0], tg[0:1], mode="sum").item(), float(g["ade_sum0"]), rtol=RTOL)
    assert np.isclose(final_displacement_error(tp[0][:, -1], tg[0:1][:, -1], mode="sum").item(),
                      float(g["fde_sum0"]), rtol=RTOL)
    # batch vs batch (train_steps.py:49-50), unknown mode -> None, shape mismatch -> AssertionError
    full = displacement_error(tp[:, 0], tg).cpu().numpy()
    assert np.allclose(full / t, g["ade"][:, 0], rtol=RTOL)
    assert displacement_error(tp[0], tg[0:1], mode="other") is None
    with pytest.raises(AssertionError):
        displacement_error(tp[0][:, :5], tg[0:1])
    with pytest.raises(Exception):
        displacement_error(tp[0].cpu(), tg[0:1].cpu())


@pytest.mark.parametrize("mode", [0, 1, 2])
def test_best_of_k_modes_vs_golden(mode, gpu_device):
    from pccs_accel.evaluation import best_of_k
    g = load_golden("g6_ade_fde")
    pred, gt = _g6_inputs(g)
    min_ade, min_fde, idx, mean_ade, mean_fde = best_of_k(dev(pred, gpu_device), dev(gt, gpu_device), mode=mode)
    assert np.array_equal(idx.cpu().numpy(), g[f"mode{mode}_idx"])  # incl. exact-tie agents 5 and 9
    assert np.allclose(min_ade.cpu().numpy(), g[f"mode{mode}_mins"][:, 0], rtol=RTOL)
    assert np.allclose(min_fde.cpu().numpy(), g[f"mode{mode}_mins"][:, 1], rtol=RTOL)
    assert np.allclose([mean_ade, mean_fde], g[f"mode{mode}_val"], rtol=RTOL)
    r = O.best_of_k(pred, gt, mode=mode)
    assert np.array_equal(idx.cpu().numpy(), r["idx"])
    assert np.allclose(min_ade.cpu().numpy(), r["min_ade"], rtol=1e-6)


@pytest.mark.parametrize("prio", ["adeprio", "fdeprio"])
def test_verify_predictions_best_of_20_f64(prio, gpu_device):
    """StimulusVerification/verify_predictions.py:238-248: float64, first 20 of 200 candidates."""
    from pccs_accel.evaluation import best_of_k
    g = load_golden("g7_verify_best_of_20")
    pred, fut = datagen.make_trajectories(int(g["seed"]), int(g["n"]), int(g["ktot"]), int(g["t"]), dtype=np.float64)
    n, t = int(g["n"]), int(g["t"])
    _, _, idx, mean_ade, mean_fde = best_of_k(dev(pred, gpu_device), dev(fut, gpu_device),
                                              mode=0 if prio == "adeprio" else 1, ade_div=1, best_of=int(g["best_of"]))
    assert np.array_equal(idx.cpu().numpy(), g[prio + "_idx"])
    assert np.isclose(mean_ade / t, float(g[prio + "_ade"]), rtol=1e-12)
    assert np.isclose(mean_fde, float(g[prio + "_fde"]), rtol=1e-12)


@pytest.mark.parametrize("n,k,t,dtype", [(1, 1, 1, np.float32), (3, 7, 5, np.float32), (1000, 100, 12, np.float32),
                                         (257, 33, 12, np.float64), (40, 300, 3, np.float32)])
def test_best_of_k_shapes(n, k, t, dtype, gpu_device):
    """Odd T (scalar loads), K > 256, single agent / candidate; all three modes vs the oracle."""
    from pccs_accel import kernels as K
    pred, gt = datagen.make_trajectories(n * 7 + k, n, k, t, dtype=dtype)
    for mode in (0, 1, 2):
        r = K.best_of_k(dev(pred, gpu_device), dev(gt, gpu_device), mode=mode)
        w = O.best_of_k(pred, gt, mode=mode)
        assert np.array_equal(r["idx"].cpu().numpy(), w["idx"])
        assert np.array_equal(r["idx2"].cpu().numpy(), w["idx2"])
        assert np.allclose(r["min_ade"].cpu().numpy(), w["min_ade"], rtol=1e-6)
        assert np.allclose(r["min_fde"].cpu().numpy(), w["min_fde"], rtol=1e-6)
        assert np.allclose(r["sums"].cpu().numpy(), [w["sum_ade"], w["sum_fde"]], rtol=1e-9)


def test_best_of_k_empty_and_bad_args(gpu_device):
    from pccs_accel import kernels as K
    pred = torch.zeros((0, 5, 12, 2), dtype=torch.float32, device=gpu_device)
    gt = torch.zeros((0, 12, 2), dtype=torch.float32, device=gpu_device)
    r = K.best_of_k(pred, gt)
    assert r["idx"].numel() == 0 and r["sums"].cpu().tolist() == [0.0, 0.0]
    with pytest.raises(ValueError):
        K.best_of_k(torch.zeros((2, 5, 12, 2), device=gpu_device), torch.zeros((2, 11, 2), device=gpu_device))
    with pytest.raises(ValueError):
        K.best_of_k(torch.zeros((2, 5, 12, 2), device=gpu_device), torch.zeros((2, 12, 2), device=gpu_device), mode=3)


# ------------------------------------------------------------------------------------------
# k-means++ seeding and the reference's call sites end to end
# ------------------------------------------------------------------------------------------
def test_kmeans_plusplus_replays_sklearn_draws(gpu_device):
    """Device k-means++ with sklearn's RandomState stream picks sklearn's centre indices."""
    from pccs_accel import seeding
    g = load_golden("g9_kmeanspp")
    X = datagen.make_encodings(int(g["seed"]), int(g["n"]), int(g["d"]), int(g["k_true"]))
    Xc = X - X.mean(axis=0)
    centers, ind = seeding.kmeans_plusplus(dev(Xc, gpu_device), int(g["K"]), np.random.RandomState(int(g["rs_seed"])))
    assert np.array_equal(ind.cpu().numpy(), g["indices"])
    assert np.array_equal(centers.cpu().numpy(), Xc[g["indices"]])


def _same_partition(a, b):
    """Label vectors equal up to a permutation of the cluster ids."""
    pairs = set(zip(a.tolist(), b.tolist()))
    return len(pairs) == len(set(a.tolist())) == len(set(b.tolist()))


def test_cluster_data_seeded_dropin(gpu_device):
    """np.random.seed(0); cluster_data(X, 20): k-means++ and 20 restarts on the GPU against the
    reference's own output (golden G2).  dtypes and the size histogram are part of the contract."""
    from pccs_accel.utils.utility import cluster_data
    g = load_golden("g2_seeded_cluster_data")
    X = datagen.make_encodings(int(g["seed"]), int(g["n"]), int(g["d"]), int(g["k_true"]))
    np.random.seed(int(g["np_seed"]))
    labels, sizes = cluster_data(X.copy(), int(g["K"]))
    assert labels.dtype == np.int32 and sizes.dtype == np.int64 and sizes.sum() == len(X)
    assert np.array_equal(np.bincount(labels, minlength=int(g["K"])), sizes)
    assert np.array_equal(labels, g["labels"])
    assert np.array_equal(sizes, g["sizes"])


def test_cluster_data_xoe_pm_dropin(gpu_device):
    """XOE variant: data[:, :obs_feat_size] *= pm before clustering (golden G8)."""
    import argparse
    from pccs_accel.utils.utility import cluster_data
    g = load_golden("g8_xoe_pm")
    X = datagen.make_encodings(int(g["seed"]), int(g["n"]), int(g["d"]), int(g["k_true"]))
    np.random.seed(int(g["np_seed"]))
    args = argparse.Namespace(pm=float(g["pm"]), obs_feat_size=int(g["obs_feat_size"]))
    labels, sizes = cluster_data(X.copy(), n_cluster=int(g["K"]), save_dir=None, args=args)
    assert np.array_equal(labels, g["labels"])
    assert np.array_equal(sizes, g["sizes"])


def test_get_memory_data_writes_reference_files(tmp_path, gpu_device):
    """get_memory_data: cache miss -> memory.npy / cluster_result.npy / size_array.npy with the
    reference's dtypes; second call hits the cache and returns the same arrays."""
    import argparse
    from pccs_accel.utils.utility import get_memory_data
    X = datagen.make_encodings(61, 1200, 96, 10)
    ds = argparse.Namespace(obs_enc=dev(X[:, :48].copy(), gpu_device), pred_enc=dev(X[:, 48:].copy(), gpu_device))
    args = argparse.Namespace(n_cluster=10, no_time_check=True)
    np.random.seed(1)
    mem, labels, sizes = get_memory_data(ds, str(tmp_path), args)
    assert mem.dtype == np.float32 and mem.shape == (1200, 96) and np.array_equal(mem, X)
    for f, dt in (("memory.npy", np.float32), ("cluster_result.npy", np.int32), ("size_array.npy", np.int64)):
        assert np.load(tmp_path / f).dtype == dt
    mem2, labels2, sizes2 = get_memory_data(None, str(tmp_path), args)   # op 6: dataset is None, cache must hit
    assert np.array_equal(labels, labels2) and np.array_equal(sizes, sizes2) and np.array_equal(mem, mem2)
    np.random.seed(1)
    ol, osz = O.cluster_data(X.copy(), 10)
    assert _same_partition(labels, ol) and np.array_equal(np.sort(sizes), np.sort(osz))


def test_kmeans_fit_rejects_non_finite_input_like_sklearn(gpu_device):
    from pccs_accel.clustering import kmeans_fit
    X = dev(datagen.make_encodings(5, 300, 96, 4), gpu_device)
    X[17, 3] = float("nan")
    with pytest.raises(ValueError, match="NaN"):
        kmeans_fit(X, 4, n_init=1)
    with pytest.raises(ValueError, match="should be >= n_clusters"):
        kmeans_fit(X[:3].contiguous(), 4, n_init=1)


@pytest.mark.parametrize("d,k", [(23, 5), (7, 3), (10, 3)])
def test_kmeans_fit_with_restarts_whose_seed_rows_start_unaligned(d, k, gpu_device):
    """K * D not a multiple of 4: restart r's seeds start r * K * D floats into the batch the seeding returns, off the
    16-byte grid the Lloyd entry asks for (the fit hands it a copy; it used to hand it the view and be refused --
    found by the sweep against sklearn, fixture G30).  Every restart against the oracle's: seeds, n_iter, labels."""
    from pccs_accel.clustering import kmeans_fit
    X = datagen.make_encodings(77, 1500, d, 4, 1500, 700)
    restarts = []
    kmeans_fit(dev(X, gpu_device), k, n_init=4, random_state=9, restarts_out=restarts)
    Xc = X - X.mean(axis=0)
    tol = float(O.tolerance(X))
    rs = np.random.RandomState(9)
    assert len(restarts) == 4
    for r in restarts:
        c0, idx = O.kmeans_plusplus(Xc, k, rs)
        ol, _, _, on = O.kmeans_single_lloyd(Xc, c0, 300, tol, sums="f64", far_order="descending")
        assert np.array_equal(r["indices"].cpu().numpy(), idx)
        assert r["n_iter"] == on and np.array_equal(r["labels"].cpu().numpy(), ol)


@pytest.mark.parametrize("n,k", [(53, 12), (1, 1), (4097, 200), (1000, 7), (0, 5)])
def test_classifier_targets_match_reference_expression(n, k, gpu_device):
    """mtp_label_targets_f32 behind classifier_targets: int64 labels and the one-hot [N, K] targets of
    PCCSNet/train_steps.py:357, :364-366 -- bit-exact against the fixture made with the reference's own torch
    expressions and against the oracle on other shapes (k % 4 != 0, empty)."""
    from pccs_accel.clustering import classifier_targets
    if (n, k) == (53, 12):
        g = load_golden("g13_classifier_targets")
        labels, want_long, want_hot = g["cluster_result"], g["cluster_results"], g["gt_prob"]
    else:
        labels = np.random.default_rng(n + k).integers(0, k, size=n).astype(np.int32)
        want_long, want_hot = O.classifier_targets(labels, k)
    lab64, hot = classifier_targets(labels, k, device=gpu_device)
    assert lab64.dtype == torch.int64 and np.array_equal(lab64.cpu().numpy(), want_long)
    assert hot.dtype == torch.float32 and tuple(hot.shape) == (n, k) and np.array_equal(hot.cpu().numpy(), want_hot)
    lab_only, none = classifier_targets(torch.from_numpy(labels).to(gpu_device), k, one_hot=False)
    assert none is None and np.array_equal(lab_only.cpu().numpy(), want_long)


def test_classifier_targets_reject_out_of_range_labels(gpu_device):
    from pccs_accel.clustering import classifier_targets
    with pytest.raises(IndexError):
        classifier_targets(np.array([0, 3, 1], dtype=np.int32), 3, device=gpu_device)
    with pytest.raises(IndexError):
        classifier_targets(np.array([0, -1], dtype=np.int32), 3, device=gpu_device)


@pytest.mark.gpu
@pytest.mark.parametrize("n", [0, 3, 64, 4097, 3 * 4096, 13291])
def test_blas_order_piecewise_entries(n, gpu_device):
    """The pieces the row-sharded seeding adds its float32 potentials from, against oracle.c's restatement of
    OpenBLAS 0.3.29's sgemv_t / sdot (what NumPy calls for sklearn's ``closest @ sample_weight`` and
    ``distance_to_candidates @ sample_weight``): 4096-point block sums + tail (mtp_blas_order_block_sums_f32), their
    sequential combination (mtp_blas_order_combine_f32), and the sdot accumulators carried over pieces of a row
    (mtp_blas_order_sdot_carry_f32) -- all bit for bit, pieces put together equal to the whole-row sums."""
    from pccs_accel import seeding
    ops = seeding.DeviceSeedOps()
    rng = np.random.default_rng(n + 5)
    b, l = 2, 3
    rows = (rng.random((b, l, n), dtype=np.float32) * 40.0).astype(np.float32)
    part = ops.block_sums(dev(rows, gpu_device)).cpu().numpy()
    for i in range(b):
        assert np.array_equal(part[i], O.sgemv_t_block_sums(rows[i]))
    comb = ops.combine(dev(part.reshape(b * l, -1), gpu_device)).cpu().numpy().reshape(b, l)
    assert np.array_equal(comb, O.combine_block_sums(part))
    if n:
        for i in range(b):
            assert np.array_equal(comb[i], O.rowsums_blas_order(rows[i]))
    if n >= 64:  # with the rows' double totals given, a row that cannot hold the smallest sum is not walked
        rows2 = rows.copy()
        rows2[:, 1, :] *= 2.0
        rows2[:, 2, :] = rows2[:, 0, ::-1]  # the same total up to rounding: both stay in
        p64 = rows2.astype(np.float64).sum(axis=2)
        part_f = ops.block_sums(dev(rows2, gpu_device), dev(p64, gpu_device), 1e-3).cpu().numpy()
        for i in range(b):
            want = O.sgemv_t_block_sums(rows2[i])
            assert np.array_equal(part_f[i, [0, 2]], want[[0, 2]]) and not part_f[i, 1, :-1].any()
    # sdot: the first restart's rows cut into three pieces at multiples of 64 (what a 4096-aligned split gives)
    x = np.ascontiguousarray(rows[:, 0, :])
    n64 = n & -64
    cuts = [0, min(n64, 64), min(n64, 4096), n]
    acc = torch.zeros((b, 64), dtype=torch.float32, device=gpu_device)
    acc_o = np.zeros((b, 64), dtype=np.float32)
    last = max([j for j in range(3) if cuts[j + 1] > cuts[j]], default=2)
    out = None
    for j in range(last + 1):
        lo, hi = cuts[j], cuts[j + 1]
        piece = np.ascontiguousarray(x[:, lo:hi])
        main_local = max(0, min(hi, n64) - lo)
        res = ops.sdot_carry(dev(piece, gpu_device).clone(), main_local, n - n64, acc, j == last)
        want = [O.sdot_carry(piece[i], main_local, n - n64, acc_o[i], j == last) for i in range(b)]
        assert np.array_equal(acc.cpu().numpy(), acc_o)
        if j == last:
            out = res.cpu().numpy()
            assert np.array_equal(out, np.array(want, dtype=np.float32))
    for i in range(b):
        assert out[i] == O.sum_sdot_order(x[i])
