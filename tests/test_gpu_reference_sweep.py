"""200 seeded random scene sets against the reference's own ``process_scene`` / ``gen_prob`` (fixture G32,
tests/golden/gen_reference_sweep.py imports them from the reference; the fixture holds digests).  The drop-ins named
like the reference's functions (pccs_accel.scene.process_scene / gen_prob, on top of mtp_scene_neighbors_f32 and
mtp_scene_soft_labels) must return the reference's neighbour dict, per-sample classes in first-seen order, their
probabilities and the dense soft labels bit for bit -- 103 111 neighbour entries over 1-4 scenes of 1-120 pedestrians,
three observation lengths, four radii, integer-built and full-mantissa positions.  G12 is the one fixed case."""
import argparse
import contextlib
import io
import os
import sys

import numpy as np
import pytest
import torch

from conftest import GOLDEN, load_golden

sys.path.insert(0, GOLDEN)
from gen_reference_sweep import crc, digests, draw_cluster_case, draw_glue_case, draw_metrics_case, draw_scene_case  # noqa: E402  (case generators and digests only)

pytestmark = pytest.mark.gpu


def test_random_scenes_bit_equal_to_the_reference(gpu_device, tmp_path):
    from pccs_accel import scene
    g = load_golden("g32_scene_sweep")
    os.makedirs(tmp_path / "ethucy" / "toy")
    bad = []
    for i, s in enumerate(g["seeds"]):
        trj, marks, obs_len, rad, k, labels = draw_scene_case(int(s))
        n = len(trj)
        assert (n, obs_len, rad, k) == (int(g["n"][i]), int(g["obs_len"][i]), float(g["rad"][i]), int(g["k"][i]))
        assert crc(trj, np.float32) == int(g["x_crc"][i]), f"case {i}: the positions are not the generator's"
        args = argparse.Namespace(obs_len=obs_len, pred_len=20 - obs_len, data_from="ethucy", rad=rad, n_cluster=k)
        train_set = argparse.Namespace(trj_data=torch.from_numpy(trj).float(), split_marks=marks)
        with contextlib.redirect_stdout(io.StringIO()):
            res = scene.process_scene("toy", train_set, args, save_dir=str(tmp_path))
        res_class, res_prob = scene.gen_prob(res, labels, str(tmp_path))
        got = digests(n, k, res, res_class, res_prob)
        diff = [key for key, v in got.items() if v != int(g[key][i])]
        # the dense labels straight from the kernel as well (get_scene_data's array, data_process.py:212-218)
        offsets, ids = scene.scene_neighbors(trj, marks, obs_len, rad)
        if crc(scene.soft_labels(offsets, ids, labels, k).cpu().numpy(), np.float64) != int(g["dense_crc"][i]):
            diff.append("dense from the kernel")
        if diff:
            bad.append((i, int(s), n, obs_len, rad, diff))
    assert not bad, f"{len(bad)} of {len(g['seeds'])} scene sets differ from the reference's: {bad[:10]}"


def test_random_candidate_sets_against_the_reference_metrics(gpu_device):
    """Fixture G33: the evaluation loop of PCCSNet/train_steps.py:214-232 around the reference's imported
    displacement_error / final_displacement_error, three modes, on 60 seeded random candidate sets with full-mantissa
    coordinates (8 755 agents, 1-40 candidates, 1-20 steps).  mtp_best_of_k_f32 must choose the reference's candidate
    for every agent in every mode (bit-exact indices) and give its ADE / FDE and the reported means to 1e-5."""
    from pccs_accel.evaluation import best_of_k
    g = load_golden("g33_metrics_sweep")
    for i, s in enumerate(g["seeds"]):
        pred, gt = draw_metrics_case(int(s))
        assert pred.shape[:3] == (int(g["n"][i]), int(g["k"][i]), int(g["t"][i]))
        assert crc(pred, np.float32) ^ crc(gt, np.float32) == int(g["x_crc"][i]), f"case {i}: not the generator's inputs"
        lo, hi = int(g["off"][i]), int(g["off"][i + 1])
        tp, tg = torch.from_numpy(pred).to(gpu_device), torch.from_numpy(gt).to(gpu_device)
        for mode in (0, 1, 2):
            min_ade, min_fde, idx, mean_ade, mean_fde = best_of_k(tp, tg, mode=mode)
            info = (i, pred.shape, mode)
            assert np.array_equal(idx.cpu().numpy(), g[f"mode{mode}_idx"][lo:hi]), info
            assert np.allclose(min_ade.cpu().numpy(), g[f"mode{mode}_mins"][lo:hi, 0], rtol=1e-5, atol=0), info
            assert np.allclose(min_fde.cpu().numpy(), g[f"mode{mode}_mins"][lo:hi, 1], rtol=1e-5, atol=0), info
            assert np.allclose([mean_ade, mean_fde], g[f"mode{mode}_val"][i], rtol=1e-5, atol=0), info


def test_cluster_data_dropin_bit_equal_to_the_reference_on_one_thread(gpu_device, monkeypatch):
    """Fixture G34: the reference's own ``np.random.seed(s); cluster_data(X, K)`` (PCCSNet/utils/utility.py:108-133)
    and the XOE copy with ``args.pm`` under ``threadpool_limits(1)``, 80 seeded random cases, 20 restarts each.  The
    drop-ins of the same names, with ``pccs_accel.clustering.SUMS = "sklearn_1t"``, return the reference's labels and
    size array bit for bit (G15 holds the six reference-sized runs; G8 one XOE case in the default mode)."""
    import argparse
    import contextlib
    import io
    from pccs_accel import clustering
    monkeypatch.setattr(clustering, "SUMS", "sklearn_1t")
    g = load_golden("g34_cluster_data_sweep")
    bad = []
    for i, s in enumerate(g["seeds"]):
        X, k, np_seed, pm, ofs = draw_cluster_case(int(s))
        assert crc(X, np.float32) == int(g["x_crc"][i]) and int(g["xoe"][i]) == int(pm is not None), i
        np.random.seed(np_seed)
        with contextlib.redirect_stdout(io.StringIO()):
            labels, sizes = clustering.cluster_data(X.copy(), k, args=None if pm is None else argparse.Namespace(pm=pm, obs_feat_size=ofs))
        if crc(labels, np.int32) != int(g["labels_crc"][i]) or crc(sizes, np.int64) != int(g["sizes_crc"][i]):
            bad.append((i, int(s), X.shape, k, pm))
    assert not bad, f"{len(bad)} of {len(g['seeds'])} cluster_data runs differ from the reference's: {bad[:10]}"


def test_modalities_and_decode_against_the_reference_functions(gpu_device):
    """Fixture G35: the reference's imported ``get_modality`` (utility.py:98-105) and ``transform_trajectory``
    (data_process.py:111-122) + gen_prediction's rotate-back (train_steps.py:253-255) on 40 seeded random cases with
    full-mantissa values.  The drop-ins: modalities to 1e-5 (double sums rounded once against torch's float32 mean;
    atol 2e-6 for the entries that cancel to ~0), the prefix sums bit for bit (4-D and 3-D, in place), the decoded
    trajectories to 1e-5."""
    from pccs_accel import kernels as K
    from pccs_accel.utils.data_process import transform_trajectory
    from pccs_accel.utils.utility import get_modality
    g = load_golden("g35_glue_sweep")
    for i, s in enumerate(g["seeds"]):
        X, labels, kk, disp, rot, last_obs = draw_glue_case(int(s))
        assert crc(X, np.float32) ^ crc(disp, np.float32) ^ crc(rot, np.float32) ^ crc(labels, np.int32) == int(g["x_crc"][i]), i
        mod = get_modality(torch.from_numpy(X), labels, kk)            # CPU tensor in, as at train_steps.py:356
        assert mod.dtype == torch.float32 and not mod.is_cuda
        assert np.allclose(mod.numpy(), g[f"mod{i}"], rtol=1e-5, atol=2e-6), (i, X.shape, kk)
        d4 = torch.from_numpy(disp.copy()).to(gpu_device)
        assert transform_trajectory(d4) is d4 and crc(d4.cpu().numpy(), np.float32) == int(g["cum4_crc"][i]), (i, disp.shape)
        d3 = torch.from_numpy(disp[:, 0].copy()).to(gpu_device)
        assert crc(transform_trajectory(d3).cpu().numpy(), np.float32) == int(g["cum3_crc"][i]), (i, disp.shape)
        dec = K.decode_trajectories(torch.from_numpy(disp).to(gpu_device), torch.from_numpy(rot).to(gpu_device),
                                    torch.from_numpy(last_obs).to(gpu_device))
        assert np.allclose(dec.cpu().numpy(), g[f"dec{i}"], rtol=1e-5, atol=1e-6), (i, disp.shape)
