"""Losses of the 3-D detection branch (TaskPrompter/detection_toolbox/det_losses.py), the part of that branch that has a
closed form and a reference host path to pin against: `FocalLoss` (sigmoid focal loss, det_losses.py:226-420; on the device the reference
calls mmcv-full 1.6.2's `sigmoid_focal_loss` extension, on the host its own `py_sigmoid_focal_loss`, :183-224) and `SmoothL1Loss`
(:102-181), with the reference's weighting / reduction rules (`weight_reduce_loss`, :28-54).  Same constructor and call signatures; the
element-wise loss, its weighted sum and the gradient run on the HIP kernels mtt_detloss_fwd / mtt_detloss_bwd (deterministic sum, the
upstream scalar read on the device: no host synchronisation).  The FCOS3D head and FPN that would feed them need mmcv / mmdet3d modules
(DCN, ConvModule, bbox coders) that are absent from this image and have no oracle: DESIGN.md §13.  Raises on CPU tensors (no fallback)."""
import torch
import torch.nn as nn

from . import ops

_KIND = dict(focal=0, smooth_l1=1)


class _DetLossFn(torch.autograd.Function):
    """-> the reduced loss (reduction 'mean' / 'sum': a scalar = sum(w * loss) * scale) or the weighted element-wise map ('none')."""

    @staticmethod
    def forward(ctx, pred, target, weight, kind, gamma, alpha, beta, reduce, scale):
        if pred.dim() != 2 or pred.dtype != torch.float32:
            raise RuntimeError("detection losses take fp32 [N, C] predictions")
        pred = pred.contiguous()
        N, Cn = pred.shape
        wmode = 0
        if weight is not None:
            weight = weight.contiguous().float()
            if weight.numel() == N and weight.shape != pred.shape:
                wmode = 1                                   # per sample (det_losses.py:331-334)
            elif weight.numel() == N * Cn:
                wmode = 2                                   # per element, possibly flattened (:336-341)
            else:
                raise ValueError(f"weight of shape {tuple(weight.shape)} fits neither [N] nor [N, C] = {tuple(pred.shape)}")
        if kind == 0:
            if target.dtype != torch.int64 or target.dim() != 1 or target.shape[0] != N:
                raise RuntimeError("focal loss takes int64 labels [N] (background = C)")     # det_losses.py:250-253
        else:
            if target.shape != pred.shape:
                raise RuntimeError("smooth L1 takes a target of the prediction's shape")      # det_losses.py:120
            target = target.float()
        target = target.contiguous()
        out = torch.empty_like(pred) if not reduce else None
        total = torch.empty(1, dtype=torch.float32, device=pred.device) if reduce else None
        kw = dict(pred=pred, target=target, weight=weight, out=out, sum=total, N=N, C=Cn, kind=kind, wmode=wmode, gamma=float(gamma),
                  alpha=float(alpha), beta=float(beta))
        ops.call("detloss_fwd", ws=ops.ws_for("detloss", pred.device, N=N, C=Cn) if reduce else None, **kw)
        ctx.save_for_backward(pred, target, weight)
        ctx.meta = (kind, wmode, float(gamma), float(alpha), float(beta), reduce, float(scale))
        return total[0] * scale if reduce else (out * scale if scale != 1.0 else out)

    @staticmethod
    def backward(ctx, g):
        pred, target, weight = ctx.saved_tensors
        kind, wmode, gamma, alpha, beta, reduce, scale = ctx.meta
        N, Cn = pred.shape
        dpred = torch.empty_like(pred)
        gs = g.reshape(1).float().contiguous() if reduce else None
        ge = None if reduce else g.contiguous().float()
        ops.call("detloss_bwd", pred=pred, target=target, weight=weight, out=None, sum=None, ws=None, N=N, C=Cn, kind=kind, wmode=wmode,
                 gamma=gamma, alpha=alpha, beta=beta, xargs=[gs, ge, scale, dpred])
        return dpred, None, None, None, None, None, None, None, None


def _reduced(pred, target, weight, kind, gamma, alpha, beta, reduction, avg_factor, loss_weight):
    """weight_reduce_loss (det_losses.py:28-54) around the element-wise kernel."""
    if reduction not in ("none", "mean", "sum"):
        raise ValueError(reduction)
    if avg_factor is None:
        if reduction == "none":
            return _DetLossFn.apply(pred, target, weight, kind, gamma, alpha, beta, False, loss_weight)
        scale = loss_weight / pred.numel() if reduction == "mean" else loss_weight
        return _DetLossFn.apply(pred, target, weight, kind, gamma, alpha, beta, True, scale)
    if reduction == "mean":
        return _DetLossFn.apply(pred, target, weight, kind, gamma, alpha, beta, True, loss_weight / float(avg_factor))
    if reduction == "none":
        return _DetLossFn.apply(pred, target, weight, kind, gamma, alpha, beta, False, loss_weight)
    raise ValueError('avg_factor can not be used with reduction="sum"')          # det_losses.py:52


class FocalLoss(nn.Module):
    """det_losses.py:347-420: sigmoid focal loss; `target` holds class indices in [0, C] with C = background."""

    def __init__(self, use_sigmoid=True, gamma=2.0, alpha=0.25, reduction='mean', loss_weight=1.0):
        super().__init__()
        assert use_sigmoid is True, 'Only sigmoid focal loss supported now.'
        self.use_sigmoid, self.gamma, self.alpha, self.reduction, self.loss_weight = use_sigmoid, gamma, alpha, reduction, loss_weight

    def forward(self, pred, target, weight=None, avg_factor=None, reduction_override=None):
        assert reduction_override in (None, 'none', 'mean', 'sum')
        reduction = reduction_override if reduction_override else self.reduction
        return _reduced(pred, target, weight, _KIND["focal"], self.gamma, self.alpha, 1.0, reduction, avg_factor, self.loss_weight)


class SmoothL1Loss(nn.Module):
    """det_losses.py:125-181."""

    def __init__(self, beta=1.0, reduction='mean', loss_weight=1.0):
        super().__init__()
        assert beta > 0
        self.beta, self.reduction, self.loss_weight = beta, reduction, loss_weight

    def forward(self, pred, target, weight=None, avg_factor=None, reduction_override=None, **kwargs):
        assert reduction_override in (None, 'none', 'mean', 'sum')
        reduction = reduction_override if reduction_override else self.reduction
        if target.numel() == 0:
            return pred.sum() * 0                                                   # det_losses.py:117-118
        shape = pred.shape
        p2 = pred.reshape(-1, shape[-1]) if pred.dim() != 2 else pred
        t2 = target.reshape(p2.shape)
        w2 = weight
        if weight is not None and weight.shape == shape and pred.dim() != 2:
            w2 = weight.reshape(p2.shape)
        out = _reduced(p2, t2, w2, _KIND["smooth_l1"], 2.0, 0.25, self.beta, reduction, avg_factor, self.loss_weight)
        return out.reshape(shape) if reduction == "none" else out
