"""The seeded whole fit at the reference's real sizes (BASELINE configs 0-2): ``np.random.seed(s);
cluster_data(X, K)`` = ``KMeans(K, n_init=20).fit`` on the train-set sizes of the five ETH/UCY leave-one-out
splits (N=73 568 eth, 76 646 hotel, 30 128 univ, 74 084 zara1, 66 976 zara2; K=200) and on N=17 970 / K=400
(SDD), against fixture G14, which tests/golden/gen_golden.py recorded from the reference itself with
sklearn's per-restart functions wrapped.

What can be asked of such a comparison, and is asserted here:
  * k-means++ is deterministic in sklearn.  With the potentials summed in the reference stack's float
    order (MTP_POT_OPENBLAS) all 20 x K seed indices must be sklearn's, exactly.
  * sklearn's Lloyd loop is NOT deterministic with more than one thread (per-thread partial sums are added
    in lock order, _k_means_lloyd.pyx:143-152): the fixture records that repeating one restart from the
    same seeds changes the labels of up to ~100 of 73 568 points and sometimes n_iter (rerun_* arrays).  So
    against the recorded run we require: the same winning restart, n_iter equal for nearly all restarts,
    label disagreement per restart of the size sklearn shows against itself, every disagreeing point as
    close to the boundary between its two centres as sklearn's own re-labelled points are.
  * The product's Lloyd loop IS deterministic (double sums).  Against the oracle run with the same
    documented variant (sums="f64") every restart must agree exactly: n_iter, labels bit for bit.
  * With ONE thread sklearn is deterministic too, and the product's sums="sklearn_1t" mode reproduces that run
    bit for bit (fixture G15, test_seeded_fit_bit_equal_to_sklearn_on_one_thread): no allowance at all.
"""
import zlib

import numpy as np
import pytest
import torch

import datagen
import kmeans_oracle as O
from conftest import load_golden

pytestmark = pytest.mark.gpu


def _rand_rows(seed, l, n):
    rs = np.random.RandomState(seed)
    return (rs.randn(l, n) * 10.0 ** rs.uniform(-3, 3, size=(l, n))).astype(np.float32)


@pytest.mark.parametrize("n", [73568, 17970, 2500, 3003, 4097, 8190, 12289, 100, 37, 5, 3])
def test_blas_order_rowsums_bit_exact(n, gpu_device):
    """mtp_blas_order_rowsums_f32 (the kernels behind MTP_POT_OPENBLAS) against oracle.c's restatement of
    the OpenBLAS sgemv_t / sdot float addition orders (itself checked against NumPy in the CPU suite):
    every row-group shape (fours, the pair, the single row), block tails, the n & 3 remainder."""
    from pccs_accel import seeding
    for l in (1, 2, 3, 4, 5, 6, 7, 8, 11, 16):
        A = _rand_rows(1000 * l + n % 997, l, n)
        got = seeding.blas_order_rowsums(torch.from_numpy(A).to(gpu_device)).cpu().numpy()
        want = O.rowsums_blas_order(A) if l > 1 else np.array([O.sum_sdot_order(A[0])], dtype=np.float32)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (n, l, got, want)
    D = (np.random.RandomState(n).rand(7, n).astype(np.float32) ** 2 * 3).astype(np.float32)  # distance-like rows
    got = seeding.blas_order_rowsums(torch.from_numpy(D).to(gpu_device)).cpu().numpy()
    assert np.array_equal(got, O.rowsums_blas_order(D))


@pytest.mark.parametrize("n,d,k,b", [(3003, 96, 8, 3), (4102, 64, 21, 4), (8190, 96, 55, 2), (2500, 128, 150, 2),
                                     (5001, 96, 404, 1)])
def test_seeding_in_openblas_order_matches_oracle(n, d, k, b, gpu_device):
    """Whole k-means++ runs with L = 4 ... 8 local trials (every row grouping of the sgemv_t order) and point
    counts with block / remainder tails, all restarts against the oracle's replay of sklearn."""
    from pccs_accel import seeding
    X = datagen.make_encodings(700 + n, n, d, 12, 1800, 500)
    Xc = np.ascontiguousarray(X - X.mean(axis=0))
    rs = np.random.RandomState(6)
    draws = [seeding.draw(rs, n, k) for _ in range(b)]
    _, idx = seeding.kmeans_plusplus_batched(torch.from_numpy(Xc).to(gpu_device), k, draws, pot_order="openblas")
    rs = np.random.RandomState(6)
    for r in range(b):
        _, oi = O.kmeans_plusplus(Xc, k, rs, pot="blas_model")
        assert np.array_equal(idx[r].cpu().numpy(), oi), (n, k, r)


@pytest.mark.parametrize("n,d", [(73568, 96), (17970, 96), (5001, 7), (1000, 128), (513, 200), (511, 16), (1, 3)])
def test_column_statistics_in_numpy_order(n, d, gpu_device):
    """mtp_col_seqsum_f32 against oracle.c's restatement of np.add.reduce(X, axis=0) (itself checked against NumPy
    in the CPU suite), and tolerance_and_mean against X.mean(axis=0) / sklearn's _tolerance: bit for bit."""
    from pccs_accel import clustering
    from pccs_accel import kernels as K
    rs = np.random.RandomState(n + d)
    X = (rs.randn(n, d) * 10.0 ** rs.uniform(-2, 2, size=(n, d))).astype(np.float32)
    Xd = torch.from_numpy(X).to(gpu_device)
    got = K.col_seqsum(Xd).cpu().numpy()
    assert np.array_equal(got.view(np.uint32), O.col_seqsum(X).view(np.uint32))
    mean = X.mean(axis=0)
    got = K.col_seqsum(Xd, torch.from_numpy(mean).to(gpu_device)).cpu().numpy()
    assert np.array_equal(got.view(np.uint32), O.col_seqsum(X, mean).view(np.uint32))
    # the chains continued over two row blocks (what the ranks of a row-sharded fit do): same bits
    a = n // 3
    part = K.col_seqsum(Xd[:a].clone())
    got = K.col_seqsum(Xd[a:].clone(), None, part).cpu().numpy()
    assert np.array_equal(got.view(np.uint32), O.col_seqsum(X).view(np.uint32))
    tol_abs, mean32 = clustering.tolerance_and_mean(Xd, 1e-4)
    assert np.array_equal(mean32.cpu().numpy().view(np.uint32), mean.view(np.uint32))
    assert tol_abs == float(O.tolerance(X, 1e-4))


def _g14_inputs(g):
    X = datagen.make_encodings(int(g["seed"]), int(g["n"]), int(g["d"]), int(g["k_true"]), int(g["spread"]),
                               int(g["noise"]))
    assert zlib.crc32(X.tobytes()) == int(g["x_crc"])
    return X


@pytest.mark.parametrize("name", ["g14_fit_sdd_n17970_k400", "g14_fit_eth_n73568_k200", "g14_fit_hotel_n76646_k200",
                                  "g14_fit_univ_n30128_k200", "g14_fit_zara1_n74084_k200", "g14_fit_zara2_n66976_k200"])
def test_seeded_fit_at_reference_size(name, gpu_device):
    from pccs_accel import clustering
    g = load_golden(name)
    X = _g14_inputs(g)
    n, K = X.shape[0], int(g["K"])
    restarts = []
    np.random.seed(int(g["np_seed"]))
    fit = clustering.kmeans_fit(torch.from_numpy(X).to(gpu_device), K, n_init=20, restarts_out=restarts)
    assert len(restarts) == 20

    # (1) seeds: exact, all restarts
    got_idx = np.stack([r["indices"].cpu().numpy() for r in restarts])
    seeds_equal = [bool(np.array_equal(got_idx[r], g["kpp_indices"][r])) for r in range(20)]
    assert all(seeds_equal), f"{name}: restarts with sklearn's seeds: {sum(seeds_equal)}/20"

    # (2) the deterministic variant: oracle Lloyd with double sums from those seeds, restart by restart, on
    # NumPy's centred points with sklearn's tolerance (the product's are the same bits, asserted first)
    tol = fit["tol_abs"]
    assert tol == float(O.tolerance(X))  # NumPy's float order for the column statistics: the same bits
    x_mean = fit["x_mean"].cpu().numpy()
    assert np.array_equal(x_mean.view(np.uint32), X.mean(axis=0).view(np.uint32))
    Xc = np.ascontiguousarray(X - x_mean)
    labels_gpu = [r["labels"].cpu().numpy() for r in restarts]
    for r in range(20):
        ol, oi, _, on = O.kmeans_single_lloyd(Xc, Xc[got_idx[r]], 300, tol, sums="f64")
        assert restarts[r]["n_iter"] == on, (name, r, restarts[r]["n_iter"], on)
        assert np.array_equal(labels_gpu[r], ol), (name, r, int((labels_gpu[r] != ol).sum()))
        assert np.isclose(restarts[r]["inertia"], oi, rtol=1e-6)

    # (3) against sklearn's recorded run
    n_iter_equal = sum(int(restarts[r]["n_iter"] == int(g["n_iter"][r])) for r in range(20))
    diffs = []
    for r in range(20):
        sizes = np.bincount(labels_gpu[r], minlength=K)
        same = zlib.crc32(np.ascontiguousarray(labels_gpu[r], dtype=np.int32).tobytes()) == int(g["labels_crc"][r])
        # the fixture keeps digests and cluster sizes per restart: half the L1 distance of the size
        # histograms bounds the number of differently labelled points from below
        diffs.append(0 if same else max(1, int(np.abs(sizes - g["restart_sizes"][r]).sum()) // 2))
        assert np.isclose(restarts[r]["inertia"], g["inertia"][r], rtol=2e-5), (name, r)
    own_spread = int(g["rerun_label_diff"].max())  # sklearn against itself
    own_iter_moves = int((g["rerun_n_iter"] != g["n_iter"][:, None]).any(axis=1).sum())
    print(f"{name}: seeds 20/20; n_iter equal to the recorded sklearn run {n_iter_equal}/20 (sklearn against itself: "
          f"{20 - own_iter_moves}/20); restarts with bit-equal labels {sum(d == 0 for d in diffs)}/20; "
          f"largest size-histogram distance {max(diffs)} points (sklearn against itself: up to {own_spread} labels)")
    assert n_iter_equal >= 20 - max(2, 2 * own_iter_moves)
    assert max(diffs) <= own_spread          # no further than sklearn moves against itself (bit-exact: the one-thread test below)

    # the winner and the final labels
    inert = np.array([r["inertia"] for r in restarts])
    assert int(np.argmin(inert)) == int(g["winner"])
    final = fit["labels"].cpu().numpy()
    assert np.array_equal(final, labels_gpu[int(g["winner"])])
    want = g["labels"].astype(np.int32)
    mism = np.nonzero(final != want)[0]
    assert mism.size <= own_spread, mism.size
    if mism.size:
        # Every point labelled differently lies about as close to the boundary between its two centres as the
        # points sklearn re-labels between two runs of its own (rerun_rel_gap).  These are not 1e-6 near-ties:
        # the loop stops on the tolerance, not at a fixed point, so one early flip moves two centres by
        # ~1/|cluster| and the runs drift apart by a few boundary points -- in sklearn as here.
        own_gap = float(g["rerun_rel_gap"].max())
        C = fit["centers"].cpu().numpy().astype(np.float64)
        x = X[mism].astype(np.float64)
        da = ((x - C[final[mism]]) ** 2).sum(axis=1)
        db = ((x - C[want[mism]]) ** 2).sum(axis=1)
        rel = np.abs(da - db) / np.maximum(da, db)
        print(f"{name}: final labels differ on {mism.size} of {n} points, relative distance gaps up to {rel.max():.2e} "
              f"(sklearn against itself: up to {own_gap:.2e})")
        assert rel.max() <= own_gap
    sizes = np.bincount(final, minlength=K)
    assert int(np.abs(sizes - g["sizes"]).sum()) <= 2 * own_spread


@pytest.mark.parametrize("name", ["g14_fit_sdd_n17970_k400", "g14_fit_eth_n73568_k200"])
def test_get_modality_at_reference_size(name, gpu_device):
    """get_modality (PCCSNet/utils/utility.py:98-105) on the reference-sized encodings with sklearn's recorded final
    labels: the segmented sums (one workgroup holds all K clusters at these sizes -- k_accumulate_all -- for K = 200;
    two cluster ranges for K = 400) against the oracle's double sums, the means against the oracle's restatement of
    the reference's per-cluster torch.mean."""
    from pccs_accel import kernels as K
    from pccs_accel.utils.utility import get_modality
    g = load_golden(name)
    X = _g14_inputs(g)
    k = int(g["K"])
    labels = g["labels"].astype(np.int32)
    Xd = torch.from_numpy(X).to(gpu_device)
    sums, counts = K.centroid_sums(Xd, torch.from_numpy(labels).to(gpu_device), k)
    ws, wc = O.accumulate_f64(X, labels, k)
    assert np.array_equal(counts.cpu().numpy(), wc) and np.array_equal(wc.astype(np.int64), g["sizes"])
    assert np.allclose(sums.cpu().numpy(), ws, rtol=1e-12, atol=1e-11)
    mod = get_modality(torch.from_numpy(X), labels, k)
    assert mod.dtype == torch.float32 and tuple(mod.shape) == (k, X.shape[1])
    assert np.allclose(mod.numpy(), O.get_modality(X, labels, k), rtol=1e-6, atol=1e-7)


G15_NAMES = ["g15_fit1t_sdd_n17970_k400", "g15_fit1t_eth_n73568_k200", "g15_fit1t_hotel_n76646_k200",
             "g15_fit1t_univ_n30128_k200", "g15_fit1t_zara1_n74084_k200", "g15_fit1t_zara2_n66976_k200"]


@pytest.mark.parametrize("name", G15_NAMES)
def test_seeded_fit_bit_equal_to_sklearn_on_one_thread(name, gpu_device):
    """The whole ``np.random.seed(s); cluster_data(X, K)`` of PCCSNet/utils/utility.py:108-133 against the reference
    stack held to ONE thread (fixture G15: sklearn is deterministic then, asserted by the generator).  With
    ``sums="sklearn_1t"`` -- float32 row-order chains for the centroid sums (_k_means_lloyd.pyx:215-218), NumPy's
    pairwise float32 total shift, float32 inertia chain -- nothing is left to tolerate: the seeds, every restart's
    n_iter / labels / inertia, the winner and the final labels are sklearn's bit for bit.  No spread allowance."""
    from pccs_accel import clustering
    g = load_golden(name)
    X = _g14_inputs(g)
    K = int(g["K"])
    restarts = []
    np.random.seed(int(g["np_seed"]))
    fit = clustering.kmeans_fit(torch.from_numpy(X).to(gpu_device), K, n_init=20, restarts_out=restarts, sums="sklearn_1t")
    assert len(restarts) == 20
    got_idx = np.stack([r["indices"].cpu().numpy() for r in restarts])
    assert np.array_equal(got_idx, g["kpp_indices"])
    for r in range(20):
        lab = np.ascontiguousarray(restarts[r]["labels"].cpu().numpy(), dtype=np.int32)
        assert restarts[r]["n_iter"] == int(g["n_iter"][r]), (name, r, restarts[r]["n_iter"], int(g["n_iter"][r]))
        assert zlib.crc32(lab.tobytes()) == int(g["labels_crc"][r]), \
            (name, r, int(np.abs(np.bincount(lab, minlength=K) - g["restart_sizes"][r]).sum()) // 2)
        assert restarts[r]["inertia"] == float(g["inertia"][r]), (name, r, restarts[r]["inertia"], float(g["inertia"][r]))
    final = fit["labels"].cpu().numpy()
    assert np.array_equal(final, g["labels"].astype(np.int32))
    assert np.array_equal(np.bincount(final, minlength=K), g["sizes"])
    assert fit["inertia"] == float(g["inertia"][int(g["winner"])]) and fit["n_iter"] == int(g["n_iter"][int(g["winner"])])
    # the restarts one at a time (mtp_lloyd_f32 instead of mtp_lloyd_batched_f32): the same bits
    if name == G15_NAMES[0]:
        one = []
        np.random.seed(int(g["np_seed"]))
        fit1 = clustering.kmeans_fit(torch.from_numpy(X).to(gpu_device), K, n_init=20, restarts_out=one, sums="sklearn_1t",
                                     batch=1)
        assert torch.equal(fit1["labels"], fit["labels"]) and torch.equal(fit1["centers"], fit["centers"])
        assert [r["n_iter"] for r in one] == [r["n_iter"] for r in restarts]
        assert [r["inertia"] for r in one] == [r["inertia"] for r in restarts]


@pytest.mark.parametrize("k", [1, 5, 7, 8, 9, 64, 127, 128, 129, 200, 255, 256, 257, 400, 1024, 3001, 8192])
def test_pairwise_sumsq_in_numpy_order(k, gpu_device):
    """mtp_pairwise_sumsq_f32 (the total shift of the sklearn_1t mode) against oracle.c's restatement of NumPy's
    pairwise float32 summation, itself pinned to ``(v ** 2).sum()`` in the CPU suite."""
    from pccs_accel._lib import check, lib, ptr, stream_ptr
    rs = np.random.RandomState(k)
    for scale in (1.0, 1e-3):
        v = (rs.rand(k) * scale * 10.0 ** rs.uniform(-2, 2, size=k)).astype(np.float32)
        out = torch.zeros(1, dtype=torch.float32, device=gpu_device)
        vd = torch.from_numpy(v).to(gpu_device)
        check(lib().mtp_pairwise_sumsq_f32(ptr(vd), k, ptr(out), stream_ptr()), "mtp_pairwise_sumsq_f32")
        got = out.cpu().numpy()[0]
        assert got.view(np.uint32) == O.pairwise_sumsq_f32(v).view(np.uint32), (k, got, O.pairwise_sumsq_f32(v))


def test_one_thread_mode_with_relocation_and_generic_d(gpu_device):
    """sums="sklearn_1t" off the beaten path, against the oracle's one-thread restatement (sums="f32", which the CPU
    suite pins to G15): an unreachable centre (relocation in float arithmetic), d = 128 and d = 64, and a feature
    size without an MFMA instantiation (d = 20: generic E-step, chain kernel with idle lanes)."""
    from pccs_accel import kernels as K
    for seed, n, d, k, bad in [(301, 3000, 96, 12, 5), (302, 2500, 128, 33, None), (303, 2047, 64, 17, 2),
                               (304, 1500, 20, 9, None), (305, 900, 200, 6, 1)]:
        X = datagen.make_encodings(seed, n, d, 10, 2000, 600)
        Xc = np.ascontiguousarray(X - X.mean(axis=0))
        C0 = Xc[datagen.make_init_indices(seed + 1, n, k)].copy()
        if bad is not None:
            C0[bad] = 50.0
        tol = float(O.tolerance(X))
        ol, oi, oc, on = O.kmeans_single_lloyd(Xc, C0, 300, tol, sums="f32", far_order="descending", inertia_order="f32_chain")
        labels, inertia, centers, n_iter = K.lloyd(torch.from_numpy(Xc).to(gpu_device), torch.from_numpy(C0).to(gpu_device),
                                                   tol_abs=tol, sums="sklearn_1t")
        assert n_iter == on, (seed, n_iter, on)
        assert np.array_equal(labels.cpu().numpy(), ol), seed
        assert np.array_equal(centers.cpu().numpy().view(np.uint32), oc.view(np.uint32)), seed
        assert inertia == oi, (seed, inertia, oi)


def test_degenerate_duplicates_match_sklearn_one_thread(gpu_device):
    """Fixture G16 (sklearn on one thread; 300 rows drawn from 5 distinct ones, K = 8): the degenerate corner of the
    loop -- clusters that stay empty, a relocation over (nearly) zero distances, the heaviest cluster's centre copied
    into the empty ones (_k_means_common.pyx:291-295), sklearn's "fewer distinct clusters" warning, k-means++ once
    every distinct point is a centre -- through the product in sums="sklearn_1t" mode: labels, centres bit for bit (the
    empty clusters' included), n_iter, inertia from the given init; the seeds while distinct points remain; the
    partition of the whole seeded fit."""
    import warnings
    from pccs_accel import clustering, seeding
    g = load_golden("g16_fewer_distinct_points_than_clusters")
    X, C0, K = g["X"], g["C0"], int(g["K"])
    Xd = torch.from_numpy(X).to(gpu_device)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        fit = clustering.kmeans_fit(Xd, K, init=C0, n_init=1, sums="sklearn_1t")
    assert any(issubclass(m.category, clustering.ConvergenceWarning) for m in w)
    assert fit["n_iter"] == int(g["n_iter"])
    assert np.array_equal(fit["labels"].cpu().numpy(), g["labels"])
    assert np.array_equal(fit["centers"].cpu().numpy().view(np.uint32), g["centers"].view(np.uint32))
    assert fit["inertia"] == float(g["inertia"])
    Xc = Xd - fit["x_mean"]
    # k-means++: sklearn's seeds while there are distinct points left to pick (the first 5 centres).  After that
    # every closest distance is (-2 x.y + |x|^2) + |y|^2 of coincident rows: cancellation noise of +-1e-16 |x|^2 whose
    # sign -- 0 after the clamp, or a tiny positive float -- depends on the accumulation order of the dgemm kernel
    # OpenBLAS picks for the matrix shape (a small-matrix kernel here, not the one of the reference's sizes).  Those
    # draws are PARITY UNPINNED; they must still be valid rows, and the partition the fit finds is the same.
    for s, seed in enumerate((0, 1, 2)):
        _, idx = seeding.kmeans_plusplus(Xc.contiguous(), K, np.random.RandomState(seed))
        idx = idx.cpu().numpy()
        assert np.array_equal(idx[:5], g["kpp_indices"][s][:5]), seed
        assert len({X[i].tobytes() for i in idx[:5]}) == 5 and ((idx >= 0) & (idx < X.shape[0])).all()
    np.random.seed(5)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        fit3 = clustering.kmeans_fit(Xd, K, n_init=3, sums="sklearn_1t")
    l3 = fit3["labels"].cpu().numpy()
    assert O.is_same_clustering(l3, g["fit_labels"], K) and O.is_same_clustering(g["fit_labels"], l3, K)
    assert fit3["inertia"] < 1e-6 and float(g["fit_inertia"]) < 1e-6   # every point sits on its centre
    # the default mode (double sums): the same labels; the populated clusters' centres to rounding
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        fit64 = clustering.kmeans_fit(Xd, K, init=C0, n_init=1)
    assert np.array_equal(fit64["labels"].cpu().numpy(), g["labels"])
    used = np.unique(g["labels"])
    assert np.allclose(fit64["centers"].cpu().numpy()[used], g["centers"][used], rtol=1e-5, atol=2e-6)


def test_extreme_cluster_counts_match_sklearn_one_thread(gpu_device):
    """Fixture G17 (sklearn on one thread): K = 1, 2, N - 1 and N clusters for N = 200 points, whole seeded fits with
    n_init = 3 -- 2 to 7 local trials in k-means++, a single cluster, every point its own centre -- in
    sums="sklearn_1t" mode: labels and centres bit for bit, n_iter, inertia."""
    from pccs_accel import clustering
    g = load_golden("g17_extreme_cluster_counts")
    Xd = torch.from_numpy(g["X"]).to(gpu_device)
    for K in g["Ks"].tolist():
        np.random.seed(int(g["np_seed"]))
        fit = clustering.kmeans_fit(Xd, K, n_init=3, sums="sklearn_1t")
        assert fit["n_iter"] == int(g[f"n_iter_{K}"]), K
        assert np.array_equal(fit["labels"].cpu().numpy(), g[f"labels_{K}"]), K
        assert np.array_equal(fit["centers"].cpu().numpy().view(np.uint32), g[f"centers_{K}"].view(np.uint32)), K
        assert fit["inertia"] == float(g[f"inertia_{K}"]), K
