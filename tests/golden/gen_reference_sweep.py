"""Generate tests/golden/g32_scene_sweep.npz: the reference's own scene filter and soft labels over seeded random scenes.

Run ONLY in the build container (imports PCCSNet/utils/data_process.py from /root/reference, never copies it):

    python tests/golden/gen_reference_sweep.py [count] [seed]            # G32, 200 cases from seed 32
    python tests/golden/gen_reference_sweep.py [count] [seed] metrics    # G33, 60 cases from seed 33
    python tests/golden/gen_reference_sweep.py [count] [seed] cluster    # G34, 80 cases from seed 34
    python tests/golden/gen_reference_sweep.py [count] [seed] glue       # G35, 40 cases from seed 35

Every case draws 1-4 scenes of 1-120 pedestrians (datagen.make_scene_trajectories, every other time jittered into
positions with full mantissas), an observation length, a radius, a cluster count and random cluster labels, runs the
reference's ``process_scene`` (:134-169) and ``gen_prob`` (:172-192), builds the dense soft labels the way
``get_scene_data`` does (:212-218) and stores the parameters, a digest of the positions and digests of the results:
neighbour offsets and ids, classes, probabilities, dense labels.  Data only.

``metrics`` (fixture G33): the evaluation loop of PCCSNet/train_steps.py:214-232 with batch size 1 around the
reference's imported ``displacement_error`` / ``final_displacement_error`` (utils/metrics.py:4-23), in its three
modes, on seeded random candidate sets with full-mantissa coordinates (1-300 agents, 1-40 candidates, 1-20 steps):
per agent the chosen candidate and its (ADE, FDE), and the means the loop reports.

``cluster`` (fixture G34): ``np.random.seed(s); cluster_data(X, K)`` of PCCSNet/utils/utility.py:108-133 and, every
other case, the XOE copy with ``args.pm`` / ``args.obs_feat_size`` (MomentaryPrediction/PCCSNet_XOE/utils/utility.py),
both imported and run under ``threadpool_limits(1)`` (20 restarts, deterministic on one thread) on seeded random
encodings: digests of the labels and of the size array.

``glue`` (fixture G35): the reference's imported ``get_modality`` (utility.py:98-105: torch.mean per cluster) on random
encodings and labels, and its imported ``transform_trajectory`` (data_process.py:111-122) on random 4-D and 3-D
displacements with full mantissas followed by gen_prediction's rotate-back with the same torch calls per agent
(train_steps.py:253-255): the modalities and decoded trajectories as float32 arrays, the prefix sums as digests
(sequential float32 adds: bit-exact).
"""
import argparse
import contextlib
import importlib
import io
import os
import sys
import tempfile
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import datagen  # noqa: E402

REF = "/root/reference"


def crc(a, dtype):
    return zlib.crc32(np.ascontiguousarray(a, dtype=dtype).tobytes())


def draw_scene_case(seed):
    """(trj float32 [n, 20, 2], split_marks, obs_len, rad, K, labels int32 [n]) of case ``seed``."""
    rs = np.random.RandomState(seed)
    sizes = [int(rs.randint(1, 120)) for _ in range(int(rs.randint(1, 5)))]
    trj = np.concatenate([datagen.make_scene_trajectories(int(rs.randint(1 << 30)), sz) for sz in sizes])
    if rs.rand() < 0.5:
        trj = trj + (rs.standard_normal(trj.shape) * 1e-2).astype(np.float32)
    trj = np.ascontiguousarray(trj.astype(np.float32))
    marks = [0] + [int(v) for v in np.cumsum(sizes)]
    obs_len = int(rs.choice([8, 2, 20]))
    rad = float(rs.choice([0.5, 1.0, 2.0, 6.0]))
    k = int(rs.randint(1, 40))
    labels = rs.randint(0, k, size=len(trj)).astype(np.int32)
    return trj, marks, obs_len, rad, k, labels


def digests(n, k, neighbours, res_class, res_prob):
    """neighbours: list of lists; res_class / res_prob: per-agent arrays -> dict of digests."""
    nb_off = np.zeros(n + 1, dtype=np.int64)
    nb_off[1:] = np.cumsum([len(neighbours[i]) for i in range(n)])
    nb_ids = np.asarray([j for i in range(n) for j in neighbours[i]], dtype=np.int64)
    dense = np.zeros((n, k)).astype(float)
    for i in range(n):
        dense[i][res_class[i]] = res_prob[i]
    return dict(nb_total=int(nb_off[-1]), nb_off_crc=crc(nb_off, np.int64), nb_ids_crc=crc(nb_ids, np.int64),
                classes_crc=crc(np.concatenate(res_class), np.int64), probs_crc=crc(np.concatenate(res_prob), np.float64),
                dense_crc=crc(dense, np.float64))


def draw_metrics_case(seed):
    """(pred float32 [n, k, t, 2], gt float32 [n, t, 2]) of G33's case ``seed``."""
    rs = np.random.RandomState(seed)
    n, k, t = int(rs.randint(1, 301)), int(rs.randint(1, 41)), int(rs.choice([12, 12, 8, 20, 5, 1]))
    pred, gt = datagen.make_trajectories(int(rs.randint(1 << 30)), n, k, t)
    scale = np.float32(rs.uniform(0.3, 3.0))
    pred = pred * scale + (rs.standard_normal(pred.shape) * 1e-2).astype(np.float32)
    gt = gt * scale + (rs.standard_normal(gt.shape) * 1e-2).astype(np.float32)
    return np.ascontiguousarray(pred.astype(np.float32)), np.ascontiguousarray(gt.astype(np.float32))


def main_metrics(count, seed0):
    import torch
    for m in [m for m in sys.modules if m == "utils" or m.startswith("utils.")]:
        del sys.modules[m]
    sys.path.insert(0, os.path.join(REF, "PCCSNet"))
    try:
        metrics = importlib.import_module("utils.metrics")
    finally:
        sys.path.remove(os.path.join(REF, "PCCSNet"))
    seeds = [int(s) for s in np.random.RandomState(seed0).randint(1 << 30, size=count)]
    out = dict(seeds=np.asarray(seeds, dtype=np.int64), n=[], k=[], t=[], x_crc=[], off=[0])
    per_mode = {m: dict(mins=[], idx=[], val=[]) for m in (0, 1, 2)}
    for s in seeds:
        pred, gt = draw_metrics_case(s)
        n, k, t = pred.shape[:3]
        tp, tg = torch.from_numpy(pred), torch.from_numpy(gt)
        for mode in (0, 1, 2):
            val = np.zeros(2).astype(float)
            mins = np.empty((n, 2), dtype=np.float32)
            idx = np.empty(n, dtype=np.int64)
            for i in range(n):  # PCCSNet/train_steps.py:214-232 with batch_size 1
                o, g = tp[i], tg[i:i + 1]
                A = metrics.displacement_error(o, g) / t
                F = metrics.final_displacement_error(o[:, -1], g[:, -1])
                if mode == 0:
                    j = torch.argmin(A)
                    a, f = A[j].item(), F[j].item()
                elif mode == 1:
                    j = torch.argmin(F)
                    a, f = A[j].item(), F[j].item()
                else:
                    j = torch.argmin(A)
                    a, f = torch.min(A).item(), torch.min(F).item()
                val += [a, f]
                mins[i] = (a, f)
                idx[i] = int(j)
            per_mode[mode]["mins"].append(mins)
            per_mode[mode]["idx"].append(idx)
            per_mode[mode]["val"].append(val / n)
        for key, v in (("n", n), ("k", k), ("t", t), ("x_crc", crc(pred, np.float32) ^ crc(gt, np.float32))):
            out[key].append(v)
        out["off"].append(out["off"][-1] + n)
    arrays = {key: np.asarray(v, dtype=np.int64) for key, v in out.items()}
    for mode in (0, 1, 2):
        arrays[f"mode{mode}_mins"] = np.concatenate(per_mode[mode]["mins"])
        arrays[f"mode{mode}_idx"] = np.concatenate(per_mode[mode]["idx"]).astype(np.int32)
        arrays[f"mode{mode}_val"] = np.stack(per_mode[mode]["val"])
    p = os.path.join(HERE, "g33_metrics_sweep.npz")
    np.savez_compressed(p, **arrays)
    print("wrote", p, os.path.getsize(p), "bytes;", int(arrays["off"][-1]), "agents")


def draw_cluster_case(seed):
    """(X float32 [n, d], K, np.random seed, pm or None, obs_feat_size) of G34's case ``seed``."""
    rs = np.random.RandomState(seed)
    n, d = int(rs.randint(300, 6000)), int(rs.choice([96, 96, 64, 128, 48, 20]))
    k = int(rs.randint(2, 41))
    X = datagen.make_encodings(int(rs.randint(1 << 30)), n, d, max(1, k // 2 + int(rs.randint(0, k))), 1500, 700)
    if rs.rand() < 0.5:
        X = X * np.float32(rs.uniform(0.3, 3.0)) + (rs.standard_normal(X.shape) * 1e-3).astype(np.float32)
    X = np.ascontiguousarray(X.astype(np.float32))
    pm = float(rs.choice([0.02, 0.5, 1.5])) if rs.rand() < 0.5 else None
    return X, k, int(rs.randint(1 << 30)), pm, d // 2


def main_cluster(count, seed0):
    from threadpoolctl import threadpool_limits
    mods = {}
    for name, root, mod in (("pccs", os.path.join(REF, "PCCSNet"), "utils.utility"),
                            ("xoe", os.path.join(REF, "MomentaryPrediction"), "PCCSNet_XOE.utils.utility")):
        for m in [m for m in sys.modules if m == "utils" or m.startswith("utils.")]:
            del sys.modules[m]
        sys.path.insert(0, root)
        try:
            mods[name] = importlib.import_module(mod)
        finally:
            sys.path.remove(root)
    seeds = [int(s) for s in np.random.RandomState(seed0).randint(1 << 30, size=count)]
    rec = {}
    with threadpool_limits(limits=1):
        for i, s in enumerate(seeds):
            X, k, np_seed, pm, ofs = draw_cluster_case(s)
            with contextlib.redirect_stdout(io.StringIO()):
                np.random.seed(np_seed)
                if pm is None:
                    labels, sizes = mods["pccs"].cluster_data(X.copy(), k)
                else:
                    labels, sizes = mods["xoe"].cluster_data(X.copy(), n_cluster=k, save_dir=None,
                                                             args=argparse.Namespace(pm=pm, obs_feat_size=ofs))
            row = dict(n=X.shape[0], d=X.shape[1], k=k, np_seed=np_seed, xoe=int(pm is not None), x_crc=crc(X, np.float32),
                       labels_crc=crc(labels, np.int32), sizes_crc=crc(sizes, np.int64))
            for key, v in row.items():
                rec.setdefault(key, []).append(v)
            if i % 10 == 0:
                print(i, X.shape, k, pm, flush=True)
    p = os.path.join(HERE, "g34_cluster_data_sweep.npz")
    np.savez_compressed(p, seeds=np.asarray(seeds, dtype=np.int64), **{key: np.asarray(v, dtype=np.int64) for key, v in rec.items()})
    print("wrote", p, os.path.getsize(p), "bytes")


def draw_glue_case(seed):
    """(X [n, d], labels int32 [n] with every cluster populated, K, disp [m, k, t, 2], rot [m, 2, 2], last_obs [m, 2])."""
    rs = np.random.RandomState(seed)
    n, d, kk = int(rs.randint(50, 4000)), int(rs.choice([96, 64, 128, 48, 20, 7])), int(rs.randint(1, 40))
    X = datagen.make_encodings(int(rs.randint(1 << 30)), n, d, max(1, kk // 2), 1500, 700)
    X = X * np.float32(rs.uniform(0.3, 3.0)) + (rs.standard_normal(X.shape) * 1e-3).astype(np.float32)
    labels = rs.randint(0, kk, size=n).astype(np.int32)
    labels[rs.choice(n, size=kk, replace=False)] = np.arange(kk, dtype=np.int32)
    m, k, t = int(rs.randint(1, 9)), int(rs.randint(1, 21)), int(rs.choice([12, 8, 5, 1]))
    disp = datagen.make_displacements(int(rs.randint(1 << 30)), m, k, t)
    disp = disp * np.float32(rs.uniform(0.3, 3.0)) + (rs.standard_normal(disp.shape) * 1e-3).astype(np.float32)
    theta = rs.uniform(-np.pi, np.pi, size=m)
    rot = np.stack([np.stack([np.cos(theta), -np.sin(theta)], -1), np.stack([np.sin(theta), np.cos(theta)], -1)], 1)
    last_obs = (rs.standard_normal((m, 2)) * 5.0).astype(np.float32)
    f32 = lambda a: np.ascontiguousarray(a.astype(np.float32))  # noqa: E731
    return f32(X), labels, kk, f32(disp), f32(rot), last_obs


def main_glue(count, seed0):
    import torch
    mods = {}
    for name in ("utils.utility", "utils.data_process"):
        for m in [m for m in sys.modules if m == "utils" or m.startswith("utils.")]:
            del sys.modules[m]
        sys.path.insert(0, os.path.join(REF, "PCCSNet"))
        try:
            mods[name] = importlib.import_module(name)
        finally:
            sys.path.remove(os.path.join(REF, "PCCSNet"))
    utility, dp = mods["utils.utility"], mods["utils.data_process"]
    seeds = [int(s) for s in np.random.RandomState(seed0).randint(1 << 30, size=count)]
    out = dict(seeds=np.asarray(seeds, dtype=np.int64))
    meta = {key: [] for key in ("x_crc", "cum4_crc", "cum3_crc")}
    for i, s in enumerate(seeds):
        X, labels, kk, disp, rot, last_obs = draw_glue_case(s)
        out[f"mod{i}"] = utility.get_modality(torch.from_numpy(X), labels, kk).numpy()
        cum4 = dp.transform_trajectory(torch.from_numpy(disp.copy())).numpy()
        cum3 = dp.transform_trajectory(torch.from_numpy(disp[:, 0].copy())).numpy()
        m, k = disp.shape[:2]
        full = []
        for a in range(m):  # train_steps.py:251-255 with batch size 1
            net_output = dp.transform_trajectory(torch.from_numpy(disp[a].copy()))
            mat = torch.from_numpy(rot[a:a + 1])
            output = torch.bmm(mat.repeat(k, 1, 1), net_output.transpose(1, 2)).transpose(1, 2)
            output += torch.from_numpy(last_obs[a:a + 1]).unsqueeze(0)
            full.append(output.numpy())
        out[f"dec{i}"] = np.stack(full)
        meta["x_crc"].append(crc(X, np.float32) ^ crc(disp, np.float32) ^ crc(rot, np.float32) ^ crc(labels, np.int32))
        meta["cum4_crc"].append(crc(cum4, np.float32))
        meta["cum3_crc"].append(crc(cum3, np.float32))
    out.update({key: np.asarray(v, dtype=np.int64) for key, v in meta.items()})
    p = os.path.join(HERE, "g35_glue_sweep.npz")
    np.savez_compressed(p, **out)
    print("wrote", p, os.path.getsize(p), "bytes")


def main():
    if len(sys.argv) > 3 and sys.argv[3] == "glue":
        return main_glue(int(sys.argv[1]), int(sys.argv[2]))
    if len(sys.argv) > 3 and sys.argv[3] == "cluster":
        return main_cluster(int(sys.argv[1]), int(sys.argv[2]))
    if len(sys.argv) > 3 and sys.argv[3] == "metrics":
        return main_metrics(int(sys.argv[1]), int(sys.argv[2]))
    import torch
    count = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 32
    for m in [m for m in sys.modules if m == "utils" or m.startswith("utils.")]:
        del sys.modules[m]
    sys.path.insert(0, os.path.join(REF, "PCCSNet"))
    try:
        dp = importlib.import_module("utils.data_process")
    finally:
        sys.path.remove(os.path.join(REF, "PCCSNet"))
    seeds = [int(s) for s in np.random.RandomState(seed0).randint(1 << 30, size=count)]
    rec = {}
    for i, s in enumerate(seeds):
        trj, marks, obs_len, rad, k, labels = draw_scene_case(s)
        n = len(trj)
        args = argparse.Namespace(obs_len=obs_len, pred_len=20 - obs_len, data_from="ethucy", rad=rad, n_cluster=k)
        train_set = argparse.Namespace(trj_data=torch.from_numpy(trj).float(), split_marks=marks)
        with tempfile.TemporaryDirectory() as tmp, contextlib.redirect_stdout(io.StringIO()):
            os.makedirs(os.path.join(tmp, "ethucy", "toy"))
            scene = dp.process_scene("toy", train_set, args, save_dir=tmp)
            res_class, res_prob = dp.gen_prob(scene, labels, tmp)
        row = dict(n=n, obs_len=obs_len, k=k, x_crc=crc(trj, np.float32), **digests(n, k, scene, res_class, res_prob))
        for key, v in row.items():
            rec.setdefault(key, []).append(v)
        rec.setdefault("rad", []).append(rad)
        if i % 10 == 0:
            print(i, n, obs_len, rad, k, row["nb_total"], flush=True)
    p = os.path.join(HERE, "g32_scene_sweep.npz")
    np.savez_compressed(p, seeds=np.asarray(seeds, dtype=np.int64),
                        **{key: np.asarray(v, dtype=np.float64 if key == "rad" else np.int64) for key, v in rec.items()})
    print("wrote", p, os.path.getsize(p), "bytes")


if __name__ == "__main__":
    main()
