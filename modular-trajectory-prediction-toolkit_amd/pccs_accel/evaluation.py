"""Best-of-K evaluation on MI355X.

``best_of_k`` is the batched core that both reference call sites reduce to:
  * PCCSNet/train_steps.py:202-239 ``evaluate`` -- per validation sample K predictions,
    ADE = displacement_error / pred_len, FDE at the last step, three selection modes
    (0 ADE-prioritised, 1 FDE-prioritised, 2 equal focus), float64 mean over samples;
  * StimulusVerification/verify_predictions.py:238-248 -- stored [N, 200, 12, 2] float64
    predictions, first ``best_of`` candidates, minADE- or minFDE-prioritised.
``evaluate`` keeps the reference's signature and print line; it batches the per-sample network
calls' outputs and scores them in one kernel launch instead of ~12 tiny launches and two
host syncs per sample.
"""
import time

import numpy as np
import torch

from . import _lib
from . import kernels as K

MODE_NAMES = {0: "ADE-prioritized", 1: "FDE-prioritized", 2: "Equal-Focus"}


def best_of_k(pred, gt, mode=0, ade_div=None, best_of=None):
    """pred [N, Ktot, T, 2], gt [N, T, 2] device tensors (float32 or float64).

    Returns (min_ade [N], min_fde [N], idx [N] int32, mean_ade, mean_fde); the means are Python
    floats of the float64 sums over agents divided by N (train_steps.py:232-233).  In mode 2
    ``idx`` is the ADE argmin.  ``ade_div`` defaults to T (``/ args.pred_len``)."""
    r = K.best_of_k(pred, gt, mode=mode, ade_div=ade_div, best_of=best_of)
    n = pred.shape[0]
    sums = r["sums"].cpu().numpy()
    return r["min_ade"], r["min_fde"], r["idx"], float(sums[0]) / max(n, 1), float(sums[1]) / max(n, 1)


def transform_trajectory(data):
    """PCCSNet/utils/data_process.py:111-122: in-place prefix sum over time ([B,T,2] or [B,K,T,2]),
    one HIP launch instead of T-1 slice adds."""
    if data.dim() not in (3, 4):
        raise TypeError
    if not (data.is_cuda and data.dtype == torch.float32 and data.is_contiguous()):
        raise _lib.MtpError("transform_trajectory: needs a contiguous float32 HIP device tensor (no CPU path)")
    view = data if data.dim() == 4 else data.unsqueeze(1)
    K.decode_trajectories(view, out=view)
    return data


def predict_topk(net, obs_enc, k, all_modalities):
    """``PCCSNet.forward`` (PCCSNet/models/components.py:116-131) for a whole batch of observation
    encodings [N, F] at once -- the reference raises NotImplementedError for bs != 1.  Same
    modules, same order: classifier -> top-k classes by descending probability -> synthesizer on
    the gathered modalities -> decoder.  Returns raw displacements [N, k, T, 2]."""
    n = obs_enc.shape[0]
    class_probs = net.classifier(obs_enc)
    prob_list = torch.argsort(class_probs, dim=1, descending=True)
    assert k <= prob_list.shape[1]
    labels = prob_list[:, :k].reshape(-1)
    expanded = obs_enc.repeat_interleave(k, dim=0)
    future_enc = net.synthesizer(expanded, labels=labels, all_modalities=all_modalities)
    out = net.decoder([expanded, future_enc])
    return out.reshape(n, k, out.shape[-2], 2)


def _collect(loader, keys):
    cols = {k: [] for k in keys}
    for data in loader:
        for k in keys:
            cols[k].append(data[k].cuda())
    return {k: torch.cat(v, dim=0) for k, v in cols.items()}


def evaluate(net, val_loader, all_modalities, dataset_name, args, batched=True):
    """Drop-in for PCCSNet/train_steps.py:202-239.  Returns np.float64[2] (ADE, FDE) of the
    last k in ``args.eval_topK`` and prints the reference's result line per k.

    The network's raw displacements for all validation samples are scored in ONE launch
    (prefix sum fused, mtp_best_of_k_disp_f32) instead of ~12 tiny launches and two host syncs per
    sample.  ``batched=True`` also runs the network on the whole set at once (predict_topk);
    ``batched=False`` keeps the reference's one-sample forward calls."""
    k_list = args.eval_topK
    assert len(k_list) > 0
    val_res = np.zeros(2).astype(float)
    for k in k_list:
        start = time.time()
        with torch.no_grad():
            if batched:
                d = _collect(val_loader, ("obs_enc", "pred"))
                disp = predict_topk(net, d["obs_enc"], k, all_modalities)
                gt = d["pred"]
            else:
                outs, gts = [], []
                for data in val_loader:
                    batch_obs_encoding, batch_pred_gt = data["obs_enc"].cuda(), data["pred"].cuda()
                    net_output = net(batch_obs_encoding, k=k, all_modalities=all_modalities)
                    assert len(net_output) == k
                    outs.append(net_output.reshape(1, k, net_output.shape[-2], 2))
                    gts.append(batch_pred_gt.reshape(1, net_output.shape[-2], 2))
                disp, gt = torch.cat(outs, dim=0), torch.cat(gts, dim=0)
        disp = disp.to(torch.float32).contiguous()
        gt = gt.to(torch.float32).contiguous()
        r = K.best_of_k_disp(disp, gt, mode=args.eval_mode, ade_div=args.pred_len, want_per_agent=False)
        sums = r["sums"].cpu().numpy()
        val_res = np.array([sums[0] / len(disp), sums[1] / len(disp)]).astype(float)
        end = time.time()
        print("==> Dataset: %s TopK: %d Mode: %s ADE: %.4f FDE: %.4f Time: %.4f"
              % (dataset_name, k, MODE_NAMES[args.eval_mode], val_res[0], val_res[1], end - start))
    return val_res


def gen_prediction(net, loader, all_modalities, dataset_name, args, batched=True, out_dir="."):
    """Drop-in for PCCSNet/train_steps.py:242-262: top-k predictions in original coordinates,
    saved as ``all_output_<dataset>_top<k>.npy`` float32 [N, k, T, 2] (the input format of
    StimulusVerification/verify_predictions.py:52-54).  Prefix sum, rotate-back and last-position
    add run as one HIP launch over all samples.  Returns the last array written."""
    import os
    k_list = args.eval_topK
    assert len(k_list) > 0
    full_output = None
    with torch.no_grad():
        for k in k_list:
            start = time.time()
            if batched:
                d = _collect(loader, ("obs_enc", "rotate_mat", "last_obs"))
                disp = predict_topk(net, d["obs_enc"], k, all_modalities)
            else:
                outs, mats, lasts = [], [], []
                for data in loader:
                    outs.append(net(data["obs_enc"].cuda(), k=k, all_modalities=all_modalities).unsqueeze(0))
                    mats.append(data["rotate_mat"].cuda())
                    lasts.append(data["last_obs"].cuda())
                disp = torch.cat(outs, dim=0)
                d = {"rotate_mat": torch.cat(mats, dim=0), "last_obs": torch.cat(lasts, dim=0)}
            disp = disp.to(torch.float32).contiguous()
            rot = d["rotate_mat"].to(torch.float32).reshape(-1, 2, 2).contiguous()
            last = d["last_obs"].to(torch.float32).reshape(-1, 2).contiguous()
            full_output = K.decode_trajectories(disp, rot, last, out=disp).cpu().numpy()
            end = time.time()
            file_name = "all_output_%s_top%d.npy" % (dataset_name, k)
            print("==> Dataset: " + dataset_name + " TopK: %d Time: %.4f FileName: %s"
                  % (k, end - start, file_name), full_output.shape)
            np.save(os.path.join(out_dir, file_name), full_output)
    return full_output
