"""Generate tests/golden/g32_scene_sweep.npz: the reference's own scene filter and soft labels over seeded random scenes.

Run ONLY in the build container (imports PCCSNet/utils/data_process.py from /root/reference, never copies it):

    python tests/golden/gen_reference_sweep.py [count] [seed]

Every case draws 1-4 scenes of 1-120 pedestrians (datagen.make_scene_trajectories, every other time jittered into
positions with full mantissas), an observation length, a radius, a cluster count and random cluster labels, runs the
reference's ``process_scene`` (:134-169) and ``gen_prob`` (:172-192), builds the dense soft labels the way
``get_scene_data`` does (:212-218) and stores the parameters, a digest of the positions and digests of the results:
neighbour offsets and ids, classes, probabilities, dense labels.  Data only.
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


def main():
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
