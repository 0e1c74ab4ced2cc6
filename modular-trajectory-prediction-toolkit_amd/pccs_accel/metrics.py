"""ADE / FDE metrics on MI355X with the reference's signatures.

Mirrors PCCSNet/utils/metrics.py:4-23 (also used by MomentaryPrediction/XOE/train.py): torch
tensors in, torch tensors out, broadcasting over dim 0 (``pred_gt`` has one row or as many as
``pred``), ``AssertionError`` on a ``shape[1:]`` mismatch, ``mode='sum'`` returns a 0-d tensor
and ``displacement_error`` returns ``None`` for any other mode, like the reference.
The reference only ever consumes the values (``.item()`` / argmin), never their gradient, so
inputs are detached.  Inputs must be HIP device tensors: there is no CPU path here.
"""
import torch

from . import kernels as K
from ._lib import MtpError


def _prep(pred, pred_gt, want_dims):
    if not (isinstance(pred, torch.Tensor) and isinstance(pred_gt, torch.Tensor)):
        raise TypeError("pred and pred_gt must be torch tensors")
    assert pred_gt.shape[1:] == pred.shape[1:]
    if pred.dim() != want_dims:
        raise ValueError(f"expected a {want_dims}-d tensor, got shape {tuple(pred.shape)}")
    if not (pred.is_cuda and pred_gt.is_cuda):
        raise MtpError("pccs_accel metrics need HIP device tensors; there is no CPU path")
    if pred_gt.shape[0] not in (1, pred.shape[0]):
        raise RuntimeError(f"The size of tensor a ({pred_gt.shape[0]}) must match the size of tensor b "
                           f"({pred.shape[0]}) at non-singleton dimension 0")
    p = pred.detach().to(torch.float32).contiguous()
    g = pred_gt.detach().to(torch.float32).contiguous()
    return p, g


def displacement_error(pred, pred_gt, mode='raw'):
    """pred [B, T, 2], pred_gt [1 or B, T, 2] -> sum_t ||gt - pred||_2 per row ([B]) or summed."""
    p, g = _prep(pred, pred_gt, 3)
    b, t, _ = p.shape
    if g.shape[0] == 1:
        ade, _ = K.ade_fde(p.view(1, b, t, 2), g.view(1, t, 2), ade_div=1.0)
    else:
        ade, _ = K.ade_fde(p.view(b, 1, t, 2), g.view(b, t, 2), ade_div=1.0)
    loss = ade.view(b)
    if mode == 'sum':
        return torch.sum(loss)
    elif mode == 'raw':
        return loss


def final_displacement_error(pred, pred_gt, mode='raw'):
    """pred [B, 2], pred_gt [1 or B, 2] -> ||gt - pred||_2 per row ([B]) or summed."""
    p, g = _prep(pred, pred_gt, 2)
    b = p.shape[0]
    if g.shape[0] == 1:
        _, fde = K.ade_fde(p.view(1, b, 1, 2), g.view(1, 1, 2), ade_div=1.0)
    else:
        _, fde = K.ade_fde(p.view(b, 1, 1, 2), g.view(b, 1, 2), ade_div=1.0)
    loss = fde.view(b)
    if mode == 'raw':
        return loss
    else:
        return torch.sum(loss)
