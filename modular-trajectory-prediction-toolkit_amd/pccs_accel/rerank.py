"""Score-sorted NMS re-ranking + best-of-K of stored prediction sets on MI355X.

Mirrors the evaluation half of StimulusVerification/verify_predictions.py (a top-level script):
    :238-248  "ADE/FDE Before Verification"  -> ``before`` of ``verify_and_evaluate``
    :251-277  per-agent score sort + greedy NMS  -> ``nms_rerank`` / ``rerank_predictions``
    :280-290  "ADE/FDE After Verification"   -> ``after``
The verifier networks that produce ``verification_scores`` (:61-235) are out of scope; scores
come in as an array, as the script's own score caches provide them.
"""
import numpy as np
import torch

from . import kernels as K


def _dev(a, device=None):
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device())
    t = a if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(a))
    return t.to(device).contiguous()


def nms_rerank(predictions, verification_scores, max_samples=200, best_of=20, bin_size=2, nms_thresh=0.3):
    """verify_predictions.py:251-277 for every agent at once.

    predictions [N, M, T, 2] (float32 / float64, NumPy or torch), verification_scores [N, M].
    Returns (order int32 [N, max_samples], n_selected int32 [N]) as device tensors:
    ``predictions[i][order[i]]`` is agent i's re-ranked candidate list."""
    pred = _dev(predictions)
    scores = _dev(verification_scores, pred.device)
    return K.nms_rerank(pred, scores, max_samples, best_of, bin_size, nms_thresh)


def rerank_predictions(predictions, order):
    """The re-ordered copy the script builds (``base_model_predictions_copy``), same type as the input."""
    pred = _dev(predictions)
    out = K.gather_candidates(pred, _dev(order, pred.device))
    return out if isinstance(predictions, torch.Tensor) else out.cpu().numpy()


def verify_and_evaluate(predictions, gt_future, verification_scores, best_of=20, max_samples=200, bin_size=2,
                        nms_thresh=0.3, fde_prioritized=False, pred_len=None, verbose=True):
    """Both evaluation blocks of the script.  gt_future [N, T, 2] = gt_trajectories[:, obs_len:].
    Returns dict(before=(ade, fde), after=(ade, fde), order, n_selected) and prints the script's
    two result lines."""
    pred = _dev(predictions)
    gt = _dev(gt_future, pred.device).to(pred.dtype)
    n, _, t, _ = pred.shape
    pred_len = t if pred_len is None else pred_len
    mode = 1 if fde_prioritized else 0
    r0 = K.best_of_k(pred, gt, mode=mode, ade_div=1.0, best_of=best_of, want_per_agent=False)
    order, n_sel = nms_rerank(pred, verification_scores, max_samples, best_of, bin_size, nms_thresh)
    r1 = K.best_of_k_reranked(pred, gt, order, mode=mode, ade_div=1.0, best_of=min(best_of, max_samples))
    s0, s1 = r0["sums"].cpu().numpy(), r1["sums"].cpu().numpy()
    before = (float(s0[0]) / n / pred_len, float(s0[1]) / n)
    after = (float(s1[0]) / n / pred_len, float(s1[1]) / n)
    if verbose:
        print('==> ADE/FDE Before Verification %.4f %.4f' % before)
        print('==> ADE/FDE After Verification %.4f %.4f' % after)
    return dict(before=before, after=after, order=order, n_selected=n_sel)
