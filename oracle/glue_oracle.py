"""CPU oracle for the steps either side of the clustering / best-of-K path.

TEST INFRASTRUCTURE ONLY -- same rule as kmeans_oracle.py: only tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg may import this module; the product never does.

Restated in NumPy / plain Python, each function citing the reference lines it follows:

* ``transform_trajectory``   /root/reference/PCCSNet/utils/data_process.py:111-122
* ``decode_predictions``     PCCSNet/train_steps.py:253-255 (prefix sum, rotate back, add last_obs)
* ``nms_rerank``             StimulusVerification/verify_predictions.py:251-277
* ``speed_and_angle``        PCCSNet/utils/data_process.py:125-131
* ``process_scene``          PCCSNet/utils/data_process.py:134-160 (returns the neighbour dict)
* ``gen_prob``               PCCSNet/utils/data_process.py:172-187
* ``scene_soft_labels``      PCCSNet/utils/data_process.py:212-218

Parity pin: tests/test_oracle_golden.py checks these against tests/golden/g10..g13*.npz, generated
by tests/golden/gen_golden.py: ``transform_trajectory``, ``process_scene``, ``gen_prob`` and
``get_scene_data`` by importing the reference's own utils/data_process.py; the bmm decode and the
NMS block by evaluating the reference's own torch / NumPy expressions (train_steps.py and
verify_predictions.py are a training driver / a top-level script and cannot be imported as
functions).
"""
import cmath

import numpy as np


# ---------------------------------------------------------------- f3: decode glue
def transform_trajectory(data):
    """data_process.py:111-122: sequential in-place prefix sum over time, in the input dtype."""
    data = np.array(data, copy=True)
    if data.ndim == 4:
        for j in range(1, data.shape[2]):
            data[:, :, j, :] = data[:, :, j, :] + data[:, :, j - 1, :]
    elif data.ndim == 3:
        for j in range(1, data.shape[1]):
            data[:, j, :] = data[:, j, :] + data[:, j - 1, :]
    else:
        raise TypeError
    return data


def decode_predictions(net_output, rot, last_obs):
    """train_steps.py:253-255 for all agents: net_output [N, k, T, 2] displacements (float32),
    rot [N, 2, 2], last_obs [N, 2] -> [N, k, T, 2] positions in original coordinates."""
    pos = transform_trajectory(np.asarray(net_output, dtype=np.float32))
    rot = np.asarray(rot, dtype=np.float32)
    out = np.empty_like(pos)
    # bmm(mat, out^T)^T: out[.., r] = mat[r, 0] * x + mat[r, 1] * y, float32
    x, y = pos[..., 0], pos[..., 1]
    for r in range(2):
        out[..., r] = rot[:, None, None, r, 0] * x + rot[:, None, None, r, 1] * y
    out += np.asarray(last_obs, dtype=np.float32)[:, None, None, :]
    return out


# ---------------------------------------------------------------- f4: NMS re-ranking
def nms_rerank(pred, scores, max_samples=200, best_of=20, bin_size=2, nms_thresh=0.3):
    """verify_predictions.py:251-277.  pred [N, Mtot, T, 2], scores [N, Mtot].

    Returns (order int32 [N, max_samples], n_selected int32 [N], reordered copy of pred):
    order[i, pos] is the index in agent i's ORIGINAL candidate list of the candidate that the
    reference's two in-place permutations leave at position pos."""
    pred = np.asarray(pred)
    scores = np.asarray(scores)
    n = pred.shape[0]
    m = max_samples
    out = np.array(pred, copy=True)
    order = np.empty((n, m), dtype=np.int32)
    n_sel = np.empty(n, dtype=np.int32)
    ramp = np.arange(m, dtype=np.float32) / np.float32(2 * m)
    for i in range(n):
        vs = scores[i][:m] // bin_size                                   # :254
        key = vs.astype(np.float32) - ramp                               # :255-256
        srt = np.argsort(-key, kind="stable")                            # :258 (descending; ties -> lower index)
        cand = out[i, :m][srt]                                           # :259-260
        selected, ignored = [0], []                                      # :263-264
        for idx in range(1, m):                                          # :265-275
            if len(selected) >= best_of:
                ignored.append(idx)
            else:
                dd = cand[selected] - cand[idx:idx + 1]
                dd = np.linalg.norm(dd, axis=-1)[..., -1]
                if np.min(dd) < nms_thresh:
                    ignored.append(idx)
                else:
                    selected.append(idx)
        new_order = selected + ignored                                   # :276
        out[i, :m] = cand[new_order]                                     # :277
        order[i] = srt[new_order]
        n_sel[i] = len(selected)
    return order, n_sel, out


# ---------------------------------------------------------------- f2: scene neighbours -> soft labels
def speed_and_angle(data, t):
    """data_process.py:125-131: polar form of (data[i, t] - data[i, 0]) via cmath (double)."""
    res = np.zeros((len(data), 2)).astype(float)
    for i in range(len(data)):
        d_x, d_y = data[i, t] - data[i, 0]
        res[i] = cmath.polar(complex(d_x, d_y))
    return res


def process_scene(trj_data, split_marks, obs_len, rad):
    """data_process.py:134-160.  trj_data float32 [N, T, 2]; returns {i: [neighbour ids]}."""
    res = {}
    for s in range(1, len(split_marks)):
        lo, hi = split_marks[s - 1], split_marks[s]
        raw = trj_data[lo:hi]
        last = raw[:, obs_len - 1]
        sa = speed_and_angle(raw, obs_len - 1)
        # vectorised over j, same arithmetic per pair as the reference's inner loop
        for i in range(len(raw)):
            # float32: sqrt(dx*dx + dy*dy), products and sum rounded separately -- what
            # np.linalg.norm(last_coord - last_coords[j]) gives for a float32 2-vector
            dist = np.linalg.norm(last[i] - last, axis=-1)
            near = dist <= rad
            near[i] = False
            smax, smin = sa[i, 0] * 1.1, sa[i, 0] * 0.9
            both_zero = (sa[:, 0] == 0) & (sa[i, 0] == 0)
            in_speed = (smin < sa[:, 0]) & (sa[:, 0] < smax)
            ad = np.abs(sa[i, 1] - sa[:, 1])
            ad = np.minimum(ad, np.pi * 2 - ad)
            ok = near & (both_zero | (~both_zero & in_speed & (ad < np.pi * 0.1)))
            res[lo + i] = [int(lo + j) for j in np.nonzero(ok)[0]]
    return res


def gen_prob(scene_dict, cluster_result):
    """data_process.py:172-187 (without the pickle): per sample the classes in first-seen order
    (own cluster first) and count / (len(neighbours) + 1)."""
    res_class, res_prob = [], []
    for i in range(len(cluster_result)):
        nb = scene_dict[i]
        tmp = {cluster_result[i]: 1}
        for c in cluster_result[nb]:
            tmp[c] = tmp.get(c, 0) + 1
        classes = np.zeros(len(tmp)).astype(int)
        prob = np.zeros(len(tmp)).astype(float)
        for idx, item in enumerate(tmp.items()):
            classes[idx] = item[0]
            prob[idx] = item[1]
        prob /= (len(nb) + 1)
        res_class.append(classes)
        res_prob.append(prob)
    return res_class, res_prob


def scene_soft_labels(res_class, res_prob, n_cluster):
    """data_process.py:212-218: dense [N, n_cluster] float64."""
    res = np.zeros((len(res_class), n_cluster)).astype(float)
    for i in range(len(res_class)):
        res[i][res_class[i]] = res_prob[i]
    return res


# ---------------------------------------------------------------- CPU baselines of the evaluation loops
def evaluate_loop_torch(pred, gt, mode=0):
    """The body of ``evaluate`` (PCCSNet/train_steps.py:214-232, with utils/metrics.py:4-23 inlined) as the
    reference runs it: one agent at a time (batch_size 1), torch ops on [k, T, 2] tensors, an ``.item()``
    pair per agent, float64 running sums.  pred [N, k, T, 2], gt [N, T, 2] torch CPU float32 tensors.
    Returns np.float64[2] = (ADE, FDE) means -- bench.py times this as the best-of-K CPU baseline."""
    import torch
    n, k, t, _ = pred.shape
    val_res = np.zeros(2).astype(float)
    for i in range(n):
        net_output, trj_gt = pred[i], gt[i:i + 1]
        ade = torch.sqrt(((trj_gt - net_output) ** 2).sum(dim=2)).sum(dim=1) / t      # displacement_error(...) / pred_len
        fde = torch.sqrt(((trj_gt[:, -1] - net_output[:, -1]) ** 2).sum(dim=1))        # final_displacement_error
        if mode == 0:
            idx = torch.argmin(ade)
            val_res += [ade[idx].item(), fde[idx].item()]
        elif mode == 1:
            idx = torch.argmin(fde)
            val_res += [ade[idx].item(), fde[idx].item()]
        else:
            val_res += [torch.min(ade).item(), torch.min(fde).item()]
    return val_res / max(n, 1)


def verify_best_of_numpy(preds, fut, best_of=20, fde_prio=False):
    """StimulusVerification/verify_predictions.py:238-248: one vectorised ``np.linalg.norm`` over the first
    ``best_of`` stored predictions, then the reference's Python loop over the agents.  preds [N, M, T, 2],
    fut [N, T, 2] (float64 in the reference).  Returns (ADE, FDE)."""
    n, _, t, _ = preds.shape
    diff = np.linalg.norm(preds[:, :best_of] - fut[:, None], axis=-1)
    tot_ade = tot_fde = 0
    for i in range(n):
        idx = np.argmin(diff[i, :, -1]) if fde_prio else np.argmin(np.sum(diff[i], axis=-1))
        tot_fde += diff[i, idx, -1]
        tot_ade += np.sum(diff[i, idx])
    return tot_ade / n / t, tot_fde / n
