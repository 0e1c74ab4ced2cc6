"""GPU parity tests of the SURVEY.md 8(f) rows through the C-ABI: decode glue (f3: prefix sum,
rotate back, fused displacement scoring, evaluate / gen_prediction drop-ins) and score-sorted NMS
re-ranking (f4), against the oracle (oracle/glue_oracle.py, pinned on the golden fixtures) and the
fixtures themselves.  Bars: integer orders / indices bit-exact, prefix sums bit-exact (same float
adds in the same order), rotated positions and ADE/FDE within 1e-5 relative."""
import argparse

import numpy as np
import pytest
import torch

import datagen
import glue_oracle as G
import kmeans_oracle as O
from conftest import load_golden

pytestmark = pytest.mark.gpu
RTOL = 1e-5


def dev(a, device):
    return torch.from_numpy(np.ascontiguousarray(a)).to(device)


# ---------------------------------------------------------------- f3
def test_transform_trajectory_bit_exact(gpu_device):
    from pccs_accel.utils.data_process import transform_trajectory
    g = load_golden("g10_decode")
    disp = datagen.make_displacements(int(g["seed"]), int(g["n"]), int(g["k"]), int(g["t"]))
    d4 = dev(disp, gpu_device)
    r4 = transform_trajectory(d4)
    assert r4 is d4 and np.array_equal(d4.cpu().numpy(), g["cum4"])     # in place, like the reference
    d3 = dev(disp[:, 0], gpu_device)
    assert np.array_equal(transform_trajectory(d3).cpu().numpy(), g["cum3"])
    with pytest.raises(TypeError):
        transform_trajectory(dev(disp[0, 0], gpu_device))


@pytest.mark.parametrize("n,k,t", [(1, 1, 1), (3, 7, 5), (257, 20, 12), (2, 3, 11), (1000, 1, 12)])
def test_decode_shapes_vs_oracle(n, k, t, gpu_device):
    from pccs_accel import kernels as K
    disp = datagen.make_displacements(100 + n, n, k, t)
    rs = np.random.RandomState(n)
    th = rs.randint(-3141, 3142, size=n) / 1000.0
    rot = np.stack([np.stack([np.cos(th), -np.sin(th)], -1), np.stack([np.sin(th), np.cos(th)], -1)], 1).astype(np.float32)
    last = (rs.randint(-40000, 40001, size=(n, 2)) / 4096.0).astype(np.float32)
    out = K.decode_trajectories(dev(disp, gpu_device), dev(rot, gpu_device), dev(last, gpu_device)).cpu().numpy()
    assert np.allclose(out, G.decode_predictions(disp, rot, last), rtol=RTOL, atol=1e-6)
    plain = K.decode_trajectories(dev(disp, gpu_device)).cpu().numpy()
    assert np.array_equal(plain, G.transform_trajectory(disp))
    # an unaligned (odd float offset) view takes the scalar path: same bits
    buf = torch.zeros(disp.size + 1, dtype=torch.float32, device=gpu_device)
    view = buf[1:].view(n, k, t, 2)
    view.copy_(dev(disp, gpu_device))
    assert np.array_equal(K.decode_trajectories(view, out=view).cpu().numpy(), plain)


def test_decode_golden(gpu_device):
    from pccs_accel import kernels as K
    g = load_golden("g10_decode")
    disp = datagen.make_displacements(int(g["seed"]), int(g["n"]), int(g["k"]), int(g["t"]))
    out = K.decode_trajectories(dev(disp, gpu_device), dev(g["rot"], gpu_device), dev(g["last_obs"], gpu_device))
    assert np.allclose(out.cpu().numpy(), g["decoded"], rtol=RTOL, atol=1e-6)


@pytest.mark.parametrize("mode", [0, 1, 2])
def test_best_of_k_on_displacements_equals_two_pass(mode, gpu_device):
    """Fused prefix sum + scoring == transform_trajectory then best_of_k, bit for bit."""
    from pccs_accel import kernels as K
    n, k, t = 513, 20, 12
    disp = datagen.make_displacements(7, n, k, t, step=300)
    _, gt = datagen.make_trajectories(8, n, 1, t, step=300)
    dd, dg = dev(disp, gpu_device), dev(gt, gpu_device)
    fused = K.best_of_k_disp(dd, dg, mode=mode)
    two = K.best_of_k(K.decode_trajectories(dd), dg, mode=mode)
    for key in ("min_ade", "min_fde", "idx", "idx2"):
        assert torch.equal(fused[key], two[key]), key
    assert torch.equal(fused["sums"], two["sums"])
    w = O.best_of_k(G.transform_trajectory(disp), gt, mode=mode)
    assert np.array_equal(fused["idx"].cpu().numpy(), w["idx"])
    assert np.isclose(float(fused["sums"][0]) / n, w["mean_ade"], rtol=RTOL)
    assert np.isclose(float(fused["sums"][1]) / n, w["mean_fde"], rtol=RTOL)


class _ToyNet(torch.nn.Module):
    """Stand-in with PCCSNet's interface (models/components.py:99-131): classifier / synthesizer /
    decoder sub-modules and a bs == 1 forward.  Weights are fixed dyadics, no training."""

    class _Cls(torch.nn.Module):
        def __init__(self, f, c):
            super().__init__()
            self.fc = torch.nn.Linear(f, c)

        def forward(self, x):
            return torch.softmax(self.fc(x), dim=1)

    class _Syn(torch.nn.Module):
        def __init__(self, f):
            super().__init__()
            self.f = f
            self.fc = torch.nn.Linear(2 * f, f)

        def forward(self, obs_encoded, labels, all_modalities):
            m = all_modalities[labels]
            return self.fc(torch.cat((torch.sigmoid(m[:, :self.f] - obs_encoded), m[:, self.f:]), dim=1))

    class _Dec(torch.nn.Module):
        def __init__(self, f, t):
            super().__init__()
            self.t = t
            self.fc = torch.nn.Linear(2 * f, 2 * t)

        def forward(self, data):
            return self.fc(torch.cat(data, 1)).reshape(len(data[0]), self.t, 2) * 0.25

    def __init__(self, f=16, c=12, t=12):
        super().__init__()
        self.classifier, self.synthesizer, self.decoder = self._Cls(f, c), self._Syn(f), self._Dec(f, t)
        g = torch.Generator().manual_seed(5)
        for p in self.parameters():
            p.data = torch.randint(-64, 65, p.shape, generator=g).float() / 128.0

    def forward(self, obs_encoded, k=None, batch_labels=None, all_modalities=None, obs_feature_len=None):
        if len(obs_encoded) != 1:
            raise NotImplementedError
        prob_list = torch.argsort(self.classifier(obs_encoded), descending=True)[0]
        expanded = obs_encoded.repeat(k, 1)
        fut = self.synthesizer(expanded, labels=prob_list[:k], all_modalities=all_modalities)
        return self.decoder([expanded, fut])


def _toy_setup(device, n=37, f=16, c=12, t=12):
    g = torch.Generator().manual_seed(11)
    net = _ToyNet(f, c, t).to(device)
    obs = torch.randint(-256, 257, (n, f), generator=g).float() / 256.0
    gt = torch.cumsum(torch.randint(-64, 65, (n, t, 2), generator=g).float() / 128.0, dim=1)
    mods = (torch.randint(-256, 257, (c, 2 * f), generator=g).float() / 256.0).to(device)
    th = torch.randint(-3141, 3142, (n,), generator=g).float() / 1000.0
    rot = torch.stack([torch.stack([torch.cos(th), -torch.sin(th)], -1), torch.stack([torch.sin(th), torch.cos(th)], -1)], 1)
    last = torch.randint(-4096, 4097, (n, 2), generator=g).float() / 256.0
    val_loader = [{"obs_enc": obs[i:i + 1], "pred": gt[i:i + 1]} for i in range(n)]
    test_loader = [{"obs_enc": obs[i:i + 1], "rotate_mat": rot[i:i + 1], "last_obs": last[i:i + 1]} for i in range(n)]
    return net, mods, val_loader, test_loader


@pytest.mark.parametrize("mode", [0, 1, 2])
def test_evaluate_dropin_matches_reference_loop(mode, gpu_device, capsys):
    """pccs_accel.evaluation.evaluate vs the loop of PCCSNet/train_steps.py:211-233 written out
    with the oracle's metrics on the same network outputs."""
    from pccs_accel.evaluation import evaluate
    net, mods, val_loader, _ = _toy_setup(gpu_device)
    args = argparse.Namespace(eval_topK=[5, 8], eval_mode=mode, pred_len=12)
    want = np.zeros(2)
    with torch.no_grad():
        for data in val_loader:
            out = net(data["obs_enc"].to(gpu_device), k=8, all_modalities=mods).cpu().numpy()
            out = G.transform_trajectory(out)
            A = O.displacement_error(out, data["pred"].numpy()) / 12
            F = O.final_displacement_error(out[:, -1], data["pred"].numpy()[:, -1])
            if mode == 0:
                want += [A[np.argmin(A)], F[np.argmin(A)]]
            elif mode == 1:
                want += [A[np.argmin(F)], F[np.argmin(F)]]
            else:
                want += [A.min(), F.min()]
    want /= len(val_loader)
    for batched in (True, False):
        got = evaluate(net, val_loader, mods, "toy", args, batched=batched)
        assert got.dtype == np.float64 and got.shape == (2,)
        assert np.allclose(got, want, rtol=RTOL)
    assert "==> Dataset: toy TopK: 8 Mode:" in capsys.readouterr().out


def test_gen_prediction_dropin(gpu_device, tmp_path):
    from pccs_accel.evaluation import gen_prediction
    net, mods, _, test_loader = _toy_setup(gpu_device)
    args = argparse.Namespace(eval_topK=[6])
    want = []
    with torch.no_grad():
        for data in test_loader:  # PCCSNet/train_steps.py:249-256 with torch on the device
            o = net(data["obs_enc"].to(gpu_device), k=6, all_modalities=mods)
            o = torch.cumsum(o, dim=1)
            o = torch.bmm(data["rotate_mat"].to(gpu_device).repeat(6, 1, 1), o.transpose(1, 2)).transpose(1, 2)
            o = o + data["last_obs"].to(gpu_device).unsqueeze(0)
            want.append(o.cpu().numpy())
    want = np.stack(want)
    for batched in (True, False):
        got = gen_prediction(net, test_loader, mods, "toy", args, batched=batched, out_dir=str(tmp_path))
        saved = np.load(tmp_path / "all_output_toy_top6.npy")
        assert saved.dtype == np.float32 and saved.shape == want.shape == (len(test_loader), 6, 12, 2)
        assert np.array_equal(saved, got) and np.allclose(got, want, rtol=RTOL, atol=1e-5)


# ---------------------------------------------------------------- f4
@pytest.mark.parametrize("tag,dtype", [("f64", np.float64), ("f32", np.float32)])
def test_nms_rerank_golden(tag, dtype, gpu_device):
    from pccs_accel import rerank
    g = load_golden("g11_nms_rerank")
    pred, fut, scores = datagen.make_candidate_sets(int(g["seed"]), int(g["n"]), int(g["m_tot"]), int(g["t"]), dtype=dtype)
    m, best_of = int(g[f"{tag}_max_samples"]), int(g["best_of"])
    order, n_sel = rerank.nms_rerank(pred, scores, max_samples=m, best_of=best_of, bin_size=int(g["bin_size"]),
                                     nms_thresh=float(g["nms_thresh"]))
    assert np.array_equal(order.cpu().numpy(), g[f"{tag}_order"])          # integer work: bit-exact
    assert np.array_equal(n_sel.cpu().numpy(), g[f"{tag}_n_sel"])
    re = rerank.rerank_predictions(pred, order)
    assert isinstance(re, np.ndarray) and re.dtype == dtype
    for i in range(len(pred)):
        assert np.array_equal(re[i, :m], pred[i][g[f"{tag}_order"][i]]) and np.array_equal(re[i, m:], pred[i, m:])
    for prio in (False, True):
        r = rerank.verify_and_evaluate(pred, fut, scores, best_of=best_of, max_samples=m, bin_size=int(g["bin_size"]),
                                       nms_thresh=float(g["nms_thresh"]), fde_prioritized=prio, verbose=False)
        p = "fdeprio" if prio else "adeprio"
        rt = 1e-12 if dtype == np.float64 else RTOL
        assert np.isclose(r["after"][0], float(g[f"{tag}_{p}_ade"]), rtol=rt)
        assert np.isclose(r["after"][1], float(g[f"{tag}_{p}_fde"]), rtol=rt)
        w = O.best_of_k(pred, fut, mode=int(prio), ade_div=1.0, best_of=best_of)
        assert np.isclose(r["before"][0] * int(g["t"]), w["mean_ade"], rtol=rt)
        assert np.isclose(r["before"][1], w["mean_fde"], rtol=rt)


@pytest.mark.parametrize("n,mtot,m,best_of,bin_size,thresh,sdtype",
                         [(1, 1, 1, 1, 2, 0.3, np.float32), (5, 64, 64, 20, 2, 0.3, np.float64),
                          (9, 200, 65, 3, 0.5, 0.05, np.float32), (4, 300, 300, 40, 3, 1.5, np.float64),
                          (3, 129, 128, 200, -2, 0.3, np.float32), (2, 1024, 1024, 20, 2, 0.2, np.float32)])
def test_nms_rerank_shapes_vs_oracle(n, mtot, m, best_of, bin_size, thresh, sdtype, gpu_device):
    """Ragged sizes (not multiples of the wave), best_of beyond the number of survivors, fractional
    and negative bins, both score dtypes; every order must equal the oracle's exactly."""
    from pccs_accel import kernels as K
    pred, fut, scores = datagen.make_candidate_sets(300 + n + mtot, n, mtot, 12, dtype=np.float32)
    scores = scores.astype(sdtype)
    order, n_sel = K.nms_rerank(dev(pred, gpu_device), dev(scores, gpu_device), m, best_of, bin_size, thresh)
    wo, ws, wout = G.nms_rerank(pred, scores, max_samples=m, best_of=best_of, bin_size=bin_size, nms_thresh=thresh)
    assert np.array_equal(order.cpu().numpy(), wo) and np.array_equal(n_sel.cpu().numpy(), ws)
    assert all(sorted(o) == list(range(m)) for o in order.cpu().numpy())   # a permutation of the first m
    k = min(best_of, m)
    r = K.best_of_k_reranked(dev(pred, gpu_device), dev(fut, gpu_device), order, mode=0, best_of=k)
    w = O.best_of_k(wout, fut, mode=0, best_of=k)
    assert np.array_equal(r["idx"].cpu().numpy(), w["idx"])
    assert np.allclose(r["min_ade"].cpu().numpy(), w["min_ade"], rtol=RTOL)


def test_rerank_argument_errors(gpu_device):
    from pccs_accel import kernels as K
    from pccs_accel._lib import MtpError
    pred = torch.zeros((2, 10, 12, 2), device=gpu_device)
    sc = torch.zeros((2, 10), device=gpu_device)
    with pytest.raises(ValueError):
        K.nms_rerank(pred, sc, 11, 20, 2, 0.3)
    with pytest.raises(MtpError):
        K.nms_rerank(pred.cpu(), sc, 10, 20, 2, 0.3)
    order, _ = K.nms_rerank(pred, sc, 10, 20, 2, 0.3)
    # identical candidates: everything after the first is within the NMS radius -> kept = [0]
    assert order.cpu().tolist() == [list(range(10))] * 2


# ---------------------------------------------------------------- f2
def _g12_inputs(g):
    return np.concatenate([datagen.make_scene_trajectories(int(s), int(n)) for s, n in zip(g["seeds"], g["sizes"])])


def _classify_neighbour_mismatches(trj, split_marks, obs_len, rad, got, want):
    """A neighbour may differ from the oracle's only where the device's hypot / atan2 rounding can
    matter: speed ratio or wrapped heading difference within 1e-12 relative of its threshold."""
    sa = G.speed_and_angle(trj, obs_len - 1)
    bad = 0
    for i in range(len(trj)):
        for j in set(got[i]) ^ set(want[i]):
            si, sj = sa[i, 0], sa[j, 0]
            ad = abs(sa[i, 1] - sa[j, 1])
            ad = min(ad, 2 * np.pi - ad)
            near = [abs(sj - si * 0.9) <= 1e-12 * si, abs(sj - si * 1.1) <= 1e-12 * si, abs(ad - np.pi * 0.1) <= 1e-12]
            bad += not any(near)
    return bad


def test_scene_neighbours_and_soft_labels_golden(gpu_device, tmp_path, monkeypatch):
    from pccs_accel import scene
    from pccs_accel.utils.data_process import gen_prob, get_scene_data, process_scene
    g = load_golden("g12_scene_soft_labels")
    trj = _g12_inputs(g)
    marks = [int(v) for v in g["split_marks"]]
    n, K = len(trj), int(g["K"])
    offsets, ids = scene.scene_neighbors(trj, marks, int(g["obs_len"]), float(g["rad"]))
    assert np.array_equal(offsets.cpu().numpy(), g["nb_off"]) and np.array_equal(ids.cpu().numpy(), g["nb_ids"])
    dense = scene.soft_labels(offsets, ids, g["labels"], K)
    assert np.array_equal(dense.cpu().numpy(), g["dense"])                  # counts / (len + 1): exact doubles
    # the drop-ins, with the reference's files
    monkeypatch.chdir(tmp_path)
    (tmp_path / "processed_scene" / "ethucy" / "toy").mkdir(parents=True)
    (tmp_path / "save").mkdir()
    args = argparse.Namespace(obs_len=int(g["obs_len"]), pred_len=12, data_from="ethucy", rad=float(g["rad"]), n_cluster=K,
                              no_time_check=True)
    train_set = argparse.Namespace(trj_data=torch.from_numpy(trj).float(), split_marks=marks)
    d = process_scene("toy", train_set, args)
    assert all(d[i] == list(g["nb_ids"][g["nb_off"][i]:g["nb_off"][i + 1]]) for i in range(n))
    import pickle
    with open(tmp_path / "processed_scene" / "ethucy" / "toy" / "scene_1.0_8.pkl", "rb") as f:
        assert pickle.load(f) == d
    rc, rp = gen_prob(d, g["labels"], str(tmp_path / "save"))
    assert np.array_equal(np.concatenate(rc), g["classes"]) and np.array_equal(np.concatenate(rp), g["probs"])
    assert [len(c) for c in rc] == list(np.diff(g["cl_off"]))
    for _ in range(2):  # second call: both caches hit
        res = get_scene_data("toy", train_set, g["labels"], str(tmp_path / "save"), args)
        assert res.dtype == np.float64 and np.array_equal(res, g["dense"])


@pytest.mark.parametrize("sizes,rad", [([1], 1.0), ([2, 0, 3], 1.0), ([65, 64, 63], 0.5), ([1500], 0.75), ([400, 900], 2.0)])
def test_scene_neighbours_shapes_vs_oracle(sizes, rad, gpu_device):
    from pccs_accel import scene
    parts = [datagen.make_scene_trajectories(500 + i + s, s, extent=5) for i, s in enumerate(sizes) if s]
    trj = np.concatenate(parts)
    marks = [0] + list(np.cumsum(sizes))
    offsets, ids = scene.scene_neighbors(trj, marks, 8, rad)
    want = G.process_scene(trj, marks, 8, rad)
    off, flat = offsets.cpu().numpy(), ids.cpu().numpy()
    got = {i: [int(v) for v in flat[off[i]:off[i + 1]]] for i in range(len(trj))}
    if got != want:
        assert _classify_neighbour_mismatches(trj, marks, 8, rad, got, want) == 0
    assert all(got[i] == sorted(got[i]) and i not in got[i] for i in got)
    labels = np.random.RandomState(len(trj)).randint(0, 37, size=len(trj)).astype(np.int32)
    dense = scene.soft_labels(offsets, ids, labels, 37).cpu().numpy()
    rc, rp = G.gen_prob(got, labels)
    assert np.array_equal(dense, G.scene_soft_labels(rc, rp, 37))
    assert np.allclose(dense.sum(axis=1), 1.0)
    with pytest.raises(IndexError):
        scene.soft_labels(offsets, ids, labels, 36 if labels.max() == 36 else int(labels.max()))


def test_speed_carries_the_bits_of_cmath_polar(gpu_device):
    """data_process.py:125-131: cmath.polar(complex(dx, dy)) -- the modulus is libm's hypot.  The kernel restates
    glibc's (>= 2.35) hypot in plain double arithmetic, so the speeds are the reference's bit for bit (the device
    library's own hypot differs in the last bit about once in 20, enough to move a neighbour whose speed ratio sits
    on the 0.9 / 1.1 boundary: found by tests/fuzz_gpu_parity.py, seed 9); the angle comes from the device's atan2
    and may differ from glibc's in the last bits (a heading difference within ~1e-15 relative of 0.1 pi)."""
    import cmath
    from pccs_accel._lib import check, lib, ptr, stream_ptr
    rs = np.random.RandomState(5)
    n = 40000
    trj = np.zeros((n, 2, 2), dtype=np.float32)
    scale = rs.choice([1e-3, 1.0, 30.0, 1e3], size=(n, 1)).astype(np.float32)
    trj[:, 1] = rs.randn(n, 2).astype(np.float32) * scale
    trj[: n // 3, 1] = np.round(trj[: n // 3, 1] * 4096) / 4096   # dyadic steps like the fixtures'
    trj[:50, 1] = 0                                              # standing still
    trj[50:100, 1, 0] = 0                                        # along an axis
    trj[:, 0] = rs.randn(n, 2).astype(np.float32)
    trj[:, 1] += trj[:, 0]
    d_trj = torch.from_numpy(trj).to(gpu_device)
    last = torch.empty((n, 2), dtype=torch.float32, device=gpu_device)
    sa = torch.empty((n, 2), dtype=torch.float64, device=gpu_device)
    check(lib().mtp_speed_angle_f32(ptr(d_trj), n, 2, 1, ptr(last), ptr(sa), stream_ptr()), "mtp_speed_angle_f32")
    got = sa.cpu().numpy()
    want = np.array([cmath.polar(complex(*(trj[i, 1] - trj[i, 0]))) for i in range(n)])
    assert np.array_equal(got[:, 0], want[:, 0])                    # the speed: every bit
    assert np.allclose(got[:, 1], want[:, 1], rtol=1e-15, atol=1e-15)
    assert np.array_equal(last.cpu().numpy(), trj[:, 1])


def test_index_inputs_are_validated_on_the_host(gpu_device):
    """Index arrays that a kernel dereferences are range-checked before the launch."""
    from pccs_accel import kernels as K
    from pccs_accel import scene
    pred = torch.zeros((2, 5, 12, 2), device=gpu_device)
    gt = torch.zeros((2, 12, 2), device=gpu_device)
    bad = torch.tensor([[0, 1, 5], [0, 1, 2]], dtype=torch.int32, device=gpu_device)
    with pytest.raises(IndexError):
        K.best_of_k_reranked(pred, gt, bad)
    with pytest.raises(IndexError):
        K.gather_candidates(pred, bad)
    off = torch.tensor([0, 1, 2], dtype=torch.int64, device=gpu_device)
    with pytest.raises(IndexError):
        scene.soft_labels(off, torch.tensor([1, 2], dtype=torch.int64, device=gpu_device), np.zeros(2, np.int32), 3)
    with pytest.raises(ValueError):
        scene.soft_labels(off, torch.tensor([1], dtype=torch.int64, device=gpu_device), np.zeros(2, np.int32), 3)
