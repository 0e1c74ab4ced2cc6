"""Modality-loss soft labels on MI355X: the neighbour search and cluster histogram that turn the
k-means labels into the classifier's default targets.

Mirrors (same names, arguments, files and return values)
    PCCSNet/utils/data_process.py:134-169  process_scene   -> {sample: [neighbour ids]} + scene_<rad>_<obs_len>.pkl
    PCCSNet/utils/data_process.py:172-192  gen_prob        -> (classes, probs) lists + probabilities.pkl
    PCCSNet/utils/data_process.py:195-220  get_scene_data  -> dense float64 [N, n_cluster]
called from PCCSNet/train_steps.py:368.  The O(n^2)-per-scene pair test and the histogram run in
libmtp_accel.so (csrc/scene.hip); this file sequences them and keeps the reference's file formats.
"""
import itertools
import os
import pickle

import numpy as np
import torch

from . import _lib
from ._lib import check, lib, ptr, stream_ptr


def _device():
    if not torch.cuda.is_available():
        raise _lib.MtpError("pccs_accel.scene needs a HIP device; there is no CPU path")
    return torch.device("cuda", torch.cuda.current_device())


def scene_neighbors(trj_data, split_marks, obs_len, rad):
    """CSR neighbour lists of every sample within its scene.

    trj_data [N, T, 2] (torch / NumPy, used as float32 like the reference's ``.float()`` data),
    split_marks: scene boundaries [0, ..., N].  Returns (offsets int64 [N+1], ids int64 [nnz]) on
    the device; sample i's neighbours are ids[offsets[i]:offsets[i+1]], ascending."""
    dev = _device()
    trj = torch.as_tensor(trj_data).to(device=dev, dtype=torch.float32).contiguous()
    if trj.dim() != 3 or trj.shape[2] != 2:
        raise ValueError(f"trj_data: expected [N, T, 2], got {tuple(trj.shape)}")
    n, tlen, _ = trj.shape
    marks = np.asarray(split_marks, dtype=np.int64)
    if marks.ndim != 1 or len(marks) < 1 or marks[0] != 0 or marks[-1] != n or np.any(np.diff(marks) < 0):
        raise ValueError("split_marks must rise from 0 to len(trj_data)")
    if not (0 <= obs_len - 1 < tlen):
        raise ValueError(f"obs_len={obs_len} does not fit trajectories of length {tlen}")
    scene_of = np.repeat(np.arange(len(marks) - 1, dtype=np.int32), np.diff(marks))
    d_marks = torch.from_numpy(marks).to(dev)
    d_scene = torch.from_numpy(scene_of).to(dev)
    last = torch.empty((n, 2), dtype=torch.float32, device=dev)
    sa = torch.empty((n, 2), dtype=torch.float64, device=dev)
    counts = torch.zeros(n, dtype=torch.int64, device=dev)
    L = lib()
    with torch.cuda.device(dev):
        check(L.mtp_speed_angle_f32(ptr(trj), n, tlen, obs_len - 1, ptr(last), ptr(sa), stream_ptr()),
              "mtp_speed_angle_f32")
        check(L.mtp_scene_neighbors_f32(ptr(last), ptr(sa), ptr(d_scene), ptr(d_marks), n, float(rad), ptr(counts), None,
                                        None, stream_ptr()), "mtp_scene_neighbors_f32")
        offsets = torch.zeros(n + 1, dtype=torch.int64, device=dev)
        torch.cumsum(counts, dim=0, out=offsets[1:])
        nnz = int(offsets[-1].item()) if n else 0
        ids = torch.empty(max(nnz, 1), dtype=torch.int64, device=dev)
        if nnz:
            check(L.mtp_scene_neighbors_f32(ptr(last), ptr(sa), ptr(d_scene), ptr(d_marks), n, float(rad), None,
                                            ptr(offsets), ptr(ids), stream_ptr()), "mtp_scene_neighbors_f32")
    return offsets, ids[:nnz]


def soft_labels(offsets, ids, cluster_result, n_cluster):
    """Dense float64 [N, n_cluster] (device): neighbour-cluster histogram incl. the sample's own
    cluster, divided by len(neighbours) + 1."""
    dev = offsets.device
    n = offsets.numel() - 1
    labels = torch.as_tensor(np.asarray(cluster_result) if not isinstance(cluster_result, torch.Tensor)
                             else cluster_result).to(device=dev, dtype=torch.int32).contiguous()
    if labels.numel() != n:
        raise ValueError("cluster_result and the neighbour lists disagree on the number of samples")
    if n and (int(labels.min()) < 0 or int(labels.max()) >= n_cluster):
        raise IndexError(f"cluster_result holds labels outside [0, {n_cluster})")
    ids = ids.to(torch.int64).contiguous()
    if ids.numel() and (int(ids.min()) < 0 or int(ids.max()) >= n):  # the kernel indexes labels with these
        raise IndexError(f"neighbour ids outside [0, {n})")
    off_h = offsets.cpu()
    if int(off_h[0]) != 0 or int(off_h[-1]) != ids.numel() or bool((off_h[1:] < off_h[:-1]).any()):
        raise ValueError("offsets must rise from 0 to len(ids)")
    dense = torch.empty((n, n_cluster), dtype=torch.float64, device=dev)
    with torch.cuda.device(dev):
        check(lib().mtp_soft_labels_f64(ptr(offsets), ptr(ids) if ids.numel() else None, ptr(labels), n, int(n_cluster),
                                        ptr(dense), stream_ptr()), "mtp_soft_labels_f64")
    return dense


def _csr_to_dict(offsets, ids):
    off = offsets.cpu().numpy()
    flat = ids.cpu().numpy().tolist()
    return {i: flat[off[i]:off[i + 1]] for i in range(len(off) - 1)}


def _dict_to_csr(scene_dict, n, device):
    lists = [scene_dict[i] for i in range(n)]
    counts = np.fromiter(map(len, lists), dtype=np.int64, count=n)
    off = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(counts, out=off[1:])
    flat = np.fromiter(itertools.chain.from_iterable(lists), dtype=np.int64, count=int(off[-1]))
    return torch.from_numpy(off).to(device), torch.from_numpy(flat).to(device)


# ------------------------------------------------------------------------------------------
# The reference's call sites
# ------------------------------------------------------------------------------------------
def process_scene(dataset_name, train_set, args, save_dir="processed_scene"):
    """PCCSNet/utils/data_process.py:134-169."""
    offsets, ids = scene_neighbors(train_set.trj_data, train_set.split_marks, args.obs_len, args.rad)
    res_dict = _csr_to_dict(offsets, ids)
    marks = list(train_set.split_marks)
    for split_idx in range(1, len(marks)):
        print("==> Scene %d with %d Trajectories Processed" % (split_idx, marks[split_idx] - marks[split_idx - 1]))
    assert len(res_dict) == len(train_set.trj_data)
    file_name = "scene_" + str(args.rad) + "_" + str(args.obs_len) + ".pkl"
    with open(os.path.join(save_dir, args.data_from, dataset_name, file_name), "wb") as f:
        pickle.dump(res_dict, f, 4)
    return res_dict


def gen_prob(scene_dict, cluster_result, save_dir):
    """PCCSNet/utils/data_process.py:172-192: per sample the classes present among itself and its
    neighbours in first-seen order (own cluster first) and their relative frequencies."""
    cluster_result = np.asarray(cluster_result)
    n = len(cluster_result)
    dev = _device()
    offsets, ids = _dict_to_csr(scene_dict, n, dev)
    k = int(cluster_result.max()) + 1 if n else 1
    dense = soft_labels(offsets, ids, cluster_result, k).cpu().numpy()
    off, flat = offsets.cpu().numpy(), ids.cpu().numpy()
    # first-seen order of the classes of [own cluster, neighbours...] for all samples in one pass (the reference
    # walks a dict per sample, data_process.py:176-186): one flat sequence ordered by (sample, position), a stable
    # sort by (sample, class) puts every pair's first occurrence first, and those first occurrences, put back in
    # sequence order, are the classes as the reference lists them
    counts = np.diff(off) + 1
    start = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(counts, out=start[1:])
    sample = np.repeat(np.arange(n, dtype=np.int64), counts)
    seq = np.empty(int(start[-1]), dtype=np.int64)
    seq[start[:-1]] = cluster_result                      # position 0: the sample's own cluster
    nb_pos = np.ones(seq.size, dtype=bool)
    nb_pos[start[:-1]] = False
    seq[nb_pos] = cluster_result[flat]                    # then its neighbours, in list order
    key = sample * k + seq
    order = np.argsort(key, kind="stable")
    sk = key[order]
    first = np.sort(order[np.concatenate(([True], sk[1:] != sk[:-1]))]) if seq.size else np.zeros(0, dtype=np.int64)
    f_sample, f_class = sample[first], seq[first].astype(int)
    cuts = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(np.bincount(f_sample, minlength=n), out=cuts[1:])
    f_prob = dense[f_sample, f_class].astype(float)
    res_class = [f_class[cuts[i]:cuts[i + 1]] for i in range(n)]  # (np.split costs five times as much)
    res_prob = [f_prob[cuts[i]:cuts[i + 1]] for i in range(n)]
    # what is left is the reference's file format: pickling 2 n small arrays takes longer than everything above
    with open(os.path.join(save_dir, "probabilities.pkl"), "wb") as f:
        pickle.dump((res_class, res_prob), f, 4)
    return res_class, res_prob


def get_scene_data(dataset_name, data_set, cluster_result, save_dir, args):
    """PCCSNet/utils/data_process.py:195-220 -> np.float64 [N, args.n_cluster].  Same caches: the
    scene pickle under processed_scene/ and probabilities.pkl in save_dir (rebuilt when stale)."""
    from .checkings import check_prob_consistency
    file_name = "scene_" + str(args.rad) + "_" + str(args.obs_len) + ".pkl"
    file_dir = os.path.join("processed_scene", args.data_from, dataset_name, file_name)
    if os.path.exists(file_dir):
        with open(file_dir, "rb") as f:
            scene_dict = pickle.load(f)
    else:
        print("==> Processing Scene...")
        scene_dict = process_scene(dataset_name, data_set, args=args)

    file_dir = os.path.join(save_dir, "probabilities.pkl")
    if os.path.exists(file_dir) and check_prob_consistency(save_dir, args):
        with open(file_dir, "rb") as f:
            gt_classes, gt_prob = pickle.load(f)
        res = np.zeros((len(gt_classes), args.n_cluster)).astype(float)
        for i in range(len(gt_classes)):
            res[i][gt_classes[i]] = gt_prob[i]
        return res
    gen_prob(scene_dict, cluster_result, save_dir)
    n = len(cluster_result)
    offsets, ids = _dict_to_csr(scene_dict, n, _device())
    return soft_labels(offsets, ids, cluster_result, args.n_cluster).cpu().numpy()
