"""TEST INFRASTRUCTURE (oracle): torch restatement of the detection branch's losses — `py_sigmoid_focal_loss` (the reference's own host
path of the sigmoid focal loss, TaskPrompter/detection_toolbox/det_losses.py:183-224; the device path is mmcv-full 1.6.2's extension with
the same closed form), `smooth_l1_loss` (:102-123), `weight_reduce_loss` (:28-54) and the `FocalLoss` / `SmoothL1Loss` modules' handling of
labels, weights, avg_factor and loss_weight (:125-181, :347-420).  Pinned against the UNMODIFIED reference file by
tests/golden/make_detloss_golden.py -> tests/golden/detloss.npz.  Only tests/ may import it."""
import torch
import torch.nn.functional as F


def weight_reduce_loss(loss, weight=None, reduction="mean", avg_factor=None):
    """det_losses.py:28-54"""
    if weight is not None:
        loss = loss * weight
    if avg_factor is None:
        return loss if reduction == "none" else (loss.mean() if reduction == "mean" else loss.sum())
    if reduction == "mean":
        return loss.sum() / avg_factor
    if reduction != "none":
        raise ValueError('avg_factor can not be used with reduction="sum"')
    return loss


def _fit_weight(weight, loss):
    """det_losses.py:207-222 / :329-343: a per-sample weight becomes a column, a flattened per-element weight is reshaped"""
    if weight is None or weight.shape == loss.shape:
        return weight
    if weight.size(0) == loss.size(0):
        return weight.view(-1, 1)
    assert weight.numel() == loss.numel()
    return weight.view(loss.size(0), -1)


def focal_loss(pred, labels, weight=None, gamma=2.0, alpha=0.25, reduction="mean", avg_factor=None, loss_weight=1.0):
    """FocalLoss.forward on the host (det_losses.py:403-420): one-hot of the labels with the background column dropped, then
    py_sigmoid_focal_loss (:183-224)."""
    C = pred.size(1)
    target = F.one_hot(labels, num_classes=C + 1)[:, :C].type_as(pred)
    p = pred.sigmoid()
    pt = (1 - p) * target + p * (1 - target)
    fw = (alpha * target + (1 - alpha) * (1 - target)) * pt.pow(gamma)
    loss = F.binary_cross_entropy_with_logits(pred, target, reduction="none") * fw
    return loss_weight * weight_reduce_loss(loss, _fit_weight(weight, loss), reduction, avg_factor)


def smooth_l1(pred, target, weight=None, beta=1.0, reduction="mean", avg_factor=None, loss_weight=1.0):
    """SmoothL1Loss.forward (det_losses.py:102-181)"""
    if target.numel() == 0:
        return pred.sum() * 0
    diff = torch.abs(pred - target)
    loss = torch.where(diff < beta, 0.5 * diff * diff / beta, diff - 0.5 * beta)
    return loss_weight * weight_reduce_loss(loss, weight, reduction, avg_factor)
