"""CPU tests of the host-side mirror of the reference interface (no GPU, no kernel calls):
cache / staleness logic of get_memory_data (PCCSNet/utils/utility.py:64-85, checkings.py:4-22),
argument validation, shard bookkeeping."""
import argparse
import os
import time

import numpy as np
import pytest
import torch


def _args(**kw):
    base = dict(n_cluster=5, no_time_check=False)
    base.update(kw)
    return argparse.Namespace(**base)


def _touch(path, t):
    with open(path, "ab"):
        pass
    os.utime(path, (t, t))


def test_get_memory_data_cache_hit_needs_no_gpu(tmp_path):
    """With valid cached artefacts nothing is recomputed (dataset may be None, as in op 6)."""
    from pccs_accel.utils.utility import get_memory_data
    mem = np.arange(40, dtype=np.float32).reshape(10, 4)
    labels = np.array([0, 1, 2, 3, 4, 0, 1, 2, 3, 4], dtype=np.int32)
    sizes = np.full(5, 2, dtype=np.int64)
    for name in ("Obs_Encoder.pth", "Pred_Encoder.pth"):
        (tmp_path / name).write_bytes(b"x")
    time.sleep(0.02)
    np.save(tmp_path / "memory.npy", mem)
    time.sleep(0.02)
    np.save(tmp_path / "cluster_result.npy", labels)
    np.save(tmp_path / "size_array.npy", sizes)
    m, l, s = get_memory_data(None, str(tmp_path), _args())
    assert np.array_equal(m, mem) and np.array_equal(l, labels) and np.array_equal(s, sizes)
    assert l.dtype == np.int32 and s.dtype == np.int64


def test_consistency_checks_follow_reference_rules(tmp_path):
    from pccs_accel.utils.checkings import check_cluster_consistency, check_memory_consistency
    for name in ("Obs_Encoder.pth", "Pred_Encoder.pth"):
        (tmp_path / name).write_bytes(b"x")
    time.sleep(0.02)
    (tmp_path / "memory.npy").write_bytes(b"x")
    time.sleep(0.02)
    (tmp_path / "cluster_result.npy").write_bytes(b"x")
    a = _args()
    assert check_memory_consistency(str(tmp_path), a)
    assert check_cluster_consistency(str(tmp_path), np.zeros(5), a)
    assert not check_cluster_consistency(str(tmp_path), np.zeros(4), a)       # wrong n_cluster
    time.sleep(0.02)
    (tmp_path / "Obs_Encoder.pth").write_bytes(b"y")                         # encoder retrained after memory
    assert not check_memory_consistency(str(tmp_path), a)
    assert check_memory_consistency(str(tmp_path), _args(no_time_check=True))
    time.sleep(0.02)
    (tmp_path / "memory.npy").write_bytes(b"y")                              # memory newer than clusters
    assert not check_cluster_consistency(str(tmp_path), np.zeros(5), a)
    assert check_cluster_consistency(str(tmp_path), np.zeros(4), _args(no_time_check=True))


def test_stale_cache_triggers_reclustering_which_needs_the_gpu(tmp_path):
    """A size_array of the wrong length invalidates the cache; the recompute path then reaches
    the HIP kernels and must fail loudly here (no GPU), never fall back to a CPU path."""
    from pccs_accel._lib import MtpError
    from pccs_accel.utils.utility import get_memory_data
    for name in ("Obs_Encoder.pth", "Pred_Encoder.pth"):
        (tmp_path / name).write_bytes(b"x")
    time.sleep(0.02)
    np.save(tmp_path / "memory.npy", np.zeros((10, 4), dtype=np.float32))
    time.sleep(0.02)
    np.save(tmp_path / "cluster_result.npy", np.zeros(10, dtype=np.int32))
    np.save(tmp_path / "size_array.npy", np.zeros(3, dtype=np.int64))
    if torch.cuda.is_available():
        pytest.skip("GPU present: the recompute path would succeed")
    with pytest.raises((MtpError, RuntimeError, AssertionError)):
        get_memory_data(None, str(tmp_path), _args(n_cluster=5, no_time_check=False))


def test_kmeans_fit_rejects_bad_input_before_any_launch():
    from pccs_accel.clustering import kmeans_fit
    with pytest.raises(TypeError):
        kmeans_fit(torch.zeros(10, 4), 2)                 # CPU tensor
    with pytest.raises(TypeError):
        kmeans_fit(np.zeros((10, 4), np.float32), 2)      # not a tensor


def test_call_site_summation_mode_is_checked_before_any_launch(monkeypatch):
    """pccs_accel.clustering.SUMS (the mode of the reference-named call sites) takes 'f64' or 'sklearn_1t' only."""
    from pccs_accel import clustering
    assert clustering.SUMS in ("f64", "sklearn_1t")
    monkeypatch.setattr(clustering, "SUMS", "float32")
    with pytest.raises(ValueError, match="SUMS"):
        clustering.k_means_cluster(np.zeros((8, 4), dtype=np.float32), 2)
    with pytest.raises(ValueError, match="SUMS"):
        clustering.cluster_data(np.zeros((8, 4), dtype=np.float32), 2)


def test_modules_mirror_reference_names():
    import pccs_accel.utils.metrics as M
    import pccs_accel.utils.utility as U
    for name in ("get_memory_data", "cluster_data", "k_means_cluster", "get_modality"):
        assert callable(getattr(U, name))
    for name in ("displacement_error", "final_displacement_error"):
        assert callable(getattr(M, name))
    from pccs_accel.evaluation import best_of_k, evaluate, gen_prediction, predict_topk  # noqa: F401
    import pccs_accel.utils.checkings as Ck
    import pccs_accel.utils.data_process as D
    for name in ("transform_trajectory", "process_scene", "gen_prob", "get_scene_data"):
        assert callable(getattr(D, name))
    for name in ("check_memory_consistency", "check_cluster_consistency", "check_prob_consistency"):
        assert callable(getattr(Ck, name))
    from pccs_accel.rerank import nms_rerank, rerank_predictions, verify_and_evaluate  # noqa: F401


def test_restart_batching_rule():
    from pccs_accel.clustering import restart_batch
    assert restart_batch(73568, 96, 200, 20) == 20        # reference scale: all restarts share the launches
    assert restart_batch(17970, 96, 400, 20) == 20
    assert restart_batch(10_000_000, 128, 1024, 20) == 1  # one restart already fills the GPU
    assert restart_batch(73568, 50, 200, 20) == 1         # generic (non-MFMA) feature size


def test_seeding_draw_order_matches_sklearn_stream():
    """seeding.draw consumes the RandomState exactly like sklearn's _kmeans_plusplus: one choice,
    then uniform(size=L) per further centre (drawn as one [K-1, L] block: same stream)."""
    import numpy as np
    from pccs_accel import seeding
    for seed, n in ((3, 500), (4, 73568), (5, 17970), (6, 3)):
        rs1, rs2 = np.random.RandomState(seed), np.random.RandomState(seed)
        w = np.ones(n, dtype=np.float32)
        for _ in range(40 if n < 1000 else 4):           # consecutive restarts on one stream
            cid, u = seeding.draw(rs1, n, 9 if n > 9 else 2)
            assert cid == rs2.choice(n, p=w / w.sum())
            for c in range(u.shape[0]):
                assert np.array_equal(u[c], rs2.uniform(size=u.shape[1]))
        assert rs1.randint(1 << 30) == rs2.randint(1 << 30)   # both streams are at the same position


def test_same_clustering_rule_matches_sklearns_loop():
    """clustering.is_same_clustering (two elementwise launches on the device) against the dict walk of
    sklearn cluster/_k_means_common.pyx:314-328 -- one-directional: l1 -> l2 must be a function."""
    from pccs_accel.clustering import is_same_clustering
    rs = np.random.RandomState(0)
    for _ in range(300):
        k, n = rs.randint(1, 6), rs.randint(0, 12)
        l1, l2 = rs.randint(0, k, n).astype(np.int32), rs.randint(0, k, n).astype(np.int32)
        mapping, want = {}, True
        for a, b in zip(l1, l2):
            if mapping.setdefault(a, b) != b:
                want = False
        assert is_same_clustering(torch.from_numpy(l1), torch.from_numpy(l2), k) == want
    assert is_same_clustering(torch.tensor([0, 1, 2, 3]), torch.tensor([0, 0, 1, 1]), 4)
    assert not is_same_clustering(torch.tensor([0, 0, 1, 1]), torch.tensor([0, 1, 2, 3]), 4)
