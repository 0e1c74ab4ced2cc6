"""240 seeded random fits against scikit-learn itself (fixture G30, tests/golden/gen_sklearn_sweep.py: sklearn 1.7.2's
``KMeans(K, n_init, random_state).fit`` held to one thread, where it is deterministic).  The fixture stores digests
only; the points are rebuilt from the case seed (their digest is asserted first).  The product's
``sums="sklearn_1t"`` mode must give sklearn's result bit for bit: iteration count, inertia, labels, cluster sizes and
the centres ``cluster_centers_`` -- for feature sizes on and off the MFMA path, integer-built and full-mantissa points,
1-5 restarts.  G15 does the same at the reference's six sizes; G31 (second test) starts from given centres.

The fits that print sklearn's ConvergenceWarning ("distinct clusters found smaller than n_clusters") print it here too."""
import os
import sys
import zlib

import numpy as np
import pytest
import torch

from conftest import GOLDEN, load_golden

sys.path.insert(0, GOLDEN)
from gen_sklearn_sweep import canonical, crc, draw_case, draw_init_case, draw_kpp_case, draw_multi_case  # noqa: E402  (the case generators only; sklearn is not imported)

pytestmark = pytest.mark.gpu


def test_random_fits_bit_equal_to_sklearn_on_one_thread(gpu_device):
    from pccs_accel import clustering
    g = load_golden("g30_sklearn_sweep")
    bad = []
    for i, s in enumerate(g["seeds"]):
        X, k, n_init, state = draw_case(int(s))
        assert (X.shape[0], X.shape[1], k, n_init) == (int(g["n"][i]), int(g["d"][i]), int(g["k"][i]), int(g["n_init"][i]))
        assert crc(X, np.float32) == int(g["x_crc"][i]), f"case {i}: the points are not the generator's"
        fit = clustering.kmeans_fit(torch.from_numpy(X).to(gpu_device), k, n_init=n_init, random_state=state, sums="sklearn_1t")
        labels = fit["labels"].cpu().numpy()
        got = dict(n_iter=fit["n_iter"] == int(g["n_iter"][i]), inertia=fit["inertia"] == float(g["inertia"][i]),
                   labels=crc(labels, np.int32) == int(g["labels_crc"][i]),
                   sizes=crc(np.bincount(labels, minlength=k), np.int64) == int(g["sizes_crc"][i]),
                   centers=crc(fit["centers"].cpu().numpy(), np.float32) == int(g["centers_crc"][i]))
        if not all(got.values()):
            bad.append((i, int(s), X.shape, k, n_init, [name for name, ok in got.items() if not ok]))
    assert not bad, f"{len(bad)} of {len(g['seeds'])} fits differ from sklearn's: {bad[:10]}"


def test_random_fits_from_given_centres_bit_equal_to_sklearn_on_one_thread(gpu_device):
    """Fixture G31: 160 ``KMeans(K, init=C0, n_init=1).fit(X)`` runs of sklearn on one thread from given centres, one
    of them out of reach in 7 cases of 10 (its cluster starts empty: sklearn's relocation to the farthest point,
    _k_means_common.pyx:165-200), over points with exact duplicate rows in 4 of 10 (ties in the E-step): bit for bit."""
    from pccs_accel import clustering
    g = load_golden("g31_sklearn_init_sweep")
    bad = []
    for i, s in enumerate(g["seeds"]):
        X, C0 = draw_init_case(int(s))
        k = C0.shape[0]
        assert (X.shape[0], X.shape[1], k) == (int(g["n"][i]), int(g["d"][i]), int(g["k"][i]))
        assert crc(X, np.float32) == int(g["x_crc"][i]), f"case {i}: the points are not the generator's"
        fit = clustering.kmeans_fit(torch.from_numpy(X).to(gpu_device), k, init=C0, n_init=1, sums="sklearn_1t")
        labels = fit["labels"].cpu().numpy()
        got = dict(n_iter=fit["n_iter"] == int(g["n_iter"][i]), inertia=fit["inertia"] == float(g["inertia"][i]),
                   labels=crc(labels, np.int32) == int(g["labels_crc"][i]),
                   sizes=crc(np.bincount(labels, minlength=k), np.int64) == int(g["sizes_crc"][i]),
                   centers=crc(fit["centers"].cpu().numpy(), np.float32) == int(g["centers_crc"][i]))
        if not all(got.values()):
            bad.append((i, int(s), X.shape, k, [name for name, ok in got.items() if not ok]))
    assert not bad, f"{len(bad)} of {len(g['seeds'])} fits differ from sklearn's: {bad[:10]}"


@pytest.mark.parametrize("sums", ["sklearn_1t", "f64"])
def test_relocation_that_takes_a_clusters_only_member(sums, gpu_device):
    """Fixture G36 (sklearn on one thread, constructed): the relocated far point was the only member of its cluster,
    which ``_average_centers`` then fills with the heaviest cluster's row as it stands in its in-place loop -- the raw
    sum if that cluster has the higher index (case a), the averaged centre otherwise (case b).  k_mstep_finish
    reproduces both (found by the sweep as an oracle infidelity: the product had it right).  1-3 iterations; the
    one-thread mode bit for bit, the default mode to 1e-5 with identical labels."""
    import warnings
    from pccs_accel import clustering
    g = load_golden("g36_relocation_takes_only_member")
    for tag in ("a", "b"):
        X, C0 = g[f"{tag}_X"], g[f"{tag}_C0"]
        for it in (1, 2, 3):
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                fit = clustering.kmeans_fit(torch.from_numpy(X).to(gpu_device), 3, init=C0, n_init=1, max_iter=it, sums=sums)
            centers, labels = fit["centers"].cpu().numpy(), fit["labels"].cpu().numpy()
            assert np.array_equal(labels, g[f"{tag}_labels{it}"]) and fit["n_iter"] == int(g[f"{tag}_n_iter{it}"]), (tag, it)
            if sums == "sklearn_1t":
                assert np.array_equal(centers.view(np.uint32), g[f"{tag}_centers{it}"].view(np.uint32)), (tag, it)
                assert fit["inertia"] == float(g[f"{tag}_inertia{it}"])
            else:
                assert np.allclose(centers, g[f"{tag}_centers{it}"], rtol=1e-5, atol=1e-6), (tag, it)


def test_random_seedings_equal_sklearns_kmeans_plusplus(gpu_device):
    """Fixture G37: sklearn's public ``kmeans_plusplus(X, K, random_state=s)`` on one thread, 120 seeded random cases --
    K 1 ... 60 and 25 cases with K = 410 / 700 / 1100 / 1250 (8 and 9 local trials), feature sizes on and off the MFMA
    path (down to D = 1), full-mantissa points.  mtp_kmeans_plusplus_f32 with the potentials in the reference stack's
    float order picks sklearn's K rows, every one of them, in every case."""
    from pccs_accel import seeding
    g = load_golden("g37_kmeanspp_sweep")
    bad = []
    for i, s in enumerate(g["seeds"]):
        X, k, state = draw_kpp_case(int(s))
        assert k == int(g["k"][i]) and crc(X, np.float32) == int(g["x_crc"][i]), f"case {i}: not the generator's points"
        centers, idx = seeding.kmeans_plusplus(torch.from_numpy(X).to(gpu_device), k, np.random.RandomState(state))
        got = idx.cpu().numpy()
        if not np.array_equal(got, g[f"idx{i}"]):
            bad.append((i, int(s), X.shape, k, int(np.argmax(got != g[f"idx{i}"]))))
        assert np.array_equal(centers.cpu().numpy(), X[got])
    assert not bad, f"{len(bad)} of {len(g['seeds'])} seedings differ from sklearn's: {bad[:10]}"


def test_several_relocations_at_once_equal_sklearn_up_to_cluster_names(gpu_device):
    """Fixture G38: 120 sklearn runs on one thread from given centres with 2 ... K-1 of them out of reach -- several
    empty clusters relocated in one iteration.  Which far point goes to which empty cluster is np.argpartition's
    business in sklearn and "descending distance" here (DESIGN.md deviation (iv)), so the names of the relocated
    clusters may differ; everything else may not: n_iter, the inertia to the last bit, the partition (labels renamed
    in order of first appearance), the sizes and the centres as a set of rows, bit for bit."""
    from pccs_accel import clustering
    g = load_golden("g38_sklearn_multi_relocation_sweep")
    bad = []
    for i, s in enumerate(g["seeds"]):
        X, C0 = draw_multi_case(int(s))
        k = C0.shape[0]
        assert crc(X, np.float32) ^ crc(C0, np.float32) == int(g["x_crc"][i]), f"case {i}: not the generator's inputs"
        fit = clustering.kmeans_fit(torch.from_numpy(X).to(gpu_device), k, init=C0, n_init=1, sums="sklearn_1t")
        lab, sizes, cen = canonical(fit["labels"].cpu().numpy(), fit["centers"].cpu().numpy())
        got = dict(n_iter=fit["n_iter"] == int(g["n_iter"][i]), inertia=fit["inertia"] == float(g["inertia"][i]),
                   partition=crc(lab, np.int32) == int(g["labels_crc"][i]), sizes=crc(sizes, np.int64) == int(g["sizes_crc"][i]),
                   centers=crc(cen, np.float32) == int(g["centers_crc"][i]))
        if not all(got.values()):
            bad.append((i, int(s), X.shape, k, [name for name, ok in got.items() if not ok]))
    assert not bad, f"{len(bad)} of {len(g['seeds'])} fits differ from sklearn's beyond the names of the clusters: {bad[:10]}"
