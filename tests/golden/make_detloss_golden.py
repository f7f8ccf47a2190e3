"""Golden fixture for the detection branch's losses, generated from the UNMODIFIED reference file
/root/reference/TaskPrompter/detection_toolbox/det_losses.py in the build container:

    python tests/golden/make_detloss_golden.py    ->  tests/golden/detloss.npz

The file imports `mmcv` and loads the compiled extension `mmcv._ext` at import time (det_losses.py:7, :229-233); both are absent here
(mmcv-full==1.6.2 / mmdet==2.28.2 are pinned in TaskPrompter/README.md:82-83, un-vendored; `from mmdet.core import bbox_overlaps` at :670
serves GIoULoss, which is not used here).  For the IMPORT to succeed empty stand-in modules are put in sys.modules — they carry no arithmetic: the cases below are evaluated through the reference's own host path (`FocalLoss.forward` on CPU
tensors takes `py_sigmoid_focal_loss`, det_losses.py:403-409; `SmoothL1Loss` is plain torch), never through the extension.  Stored per
case: inputs, loss value(s) and d loss / d pred from the reference's autograd."""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("MTT_REFERENCE_ROOT", "/root/reference")

# (name, kind, N, C, weight mode, reduction, avg_factor, loss_weight, extra)
CASES = [
    ("focal_mean", "focal", 37, 10, None, "mean", None, 1.0, dict(gamma=2.0, alpha=0.25)),
    ("focal_avg", "focal", 64, 10, "sample", "mean", 23.0, 1.0, dict(gamma=2.0, alpha=0.25)),          # FCOS3D's call: avg_factor = num_pos
    ("focal_sum_w", "focal", 50, 3, "elem", "sum", None, 0.7, dict(gamma=1.5, alpha=0.4)),
    ("focal_none", "focal", 33, 5, "sample", "none", None, 2.0, dict(gamma=2.0, alpha=0.25)),
    ("focal_big", "focal", 29, 4, None, "mean", None, 1.0, dict(gamma=2.0, alpha=0.25, logit_scale=30.0)),   # saturated logits
    ("sl1_mean", "sl1", 41, 7, None, "mean", None, 1.0, dict(beta=1.0 / 9.0)),                                 # det_head_params.py:50
    ("sl1_avg_w", "sl1", 48, 2, "elem", "mean", 17.0, 1.0, dict(beta=1.0 / 9.0)),
    ("sl1_none", "sl1", 19, 9, "elem", "none", None, 0.5, dict(beta=0.5)),
    ("sl1_sum", "sl1", 25, 3, None, "sum", None, 1.0, dict(beta=1.0)),
]


def inputs(name, kind, N, C, wmode, extra):
    g = torch.Generator().manual_seed(abs(hash(name)) % (2 ** 31) if False else sum(ord(c) for c in name))
    pred = torch.randn(N, C, generator=g) * extra.get("logit_scale", 2.0)
    if kind == "focal":
        target = torch.randint(0, C + 1, (N,), generator=g)             # C = background
    else:
        target = pred + torch.randn(N, C, generator=g) * 0.3            # differences on both sides of beta
    weight = None
    if wmode == "sample":
        weight = torch.rand(N, generator=g) + 0.1
    elif wmode == "elem":
        weight = torch.rand(N, C, generator=g) + 0.1
    return pred, target, weight


def main():
    for m in ("mmcv", "mmcv._ext", "mmdet", "mmdet.core"):          # import-only stand-ins (see the module docstring)
        mod = types.ModuleType(m)
        for f in ("sigmoid_focal_loss_forward", "sigmoid_focal_loss_backward", "softmax_focal_loss_forward", "softmax_focal_loss_backward",
                  "bbox_overlaps"):
            setattr(mod, f, None)
        mod.jit = lambda **kw: (lambda fn: fn)                      # the decorator of giou_loss (det_losses.py:671), an identity in mmcv 1.6.2 outside parrots
        sys.modules[m] = mod
    sys.path.insert(0, os.path.join(REF, "TaskPrompter"))
    from detection_toolbox import det_losses as ref
    out = {}
    for name, kind, N, C, wmode, reduction, avg, lw, extra in CASES:
        pred, target, weight = inputs(name, kind, N, C, wmode, extra)
        pr = pred.clone().requires_grad_(True)
        if kind == "focal":
            crit = ref.FocalLoss(use_sigmoid=True, gamma=extra["gamma"], alpha=extra["alpha"], reduction=reduction, loss_weight=lw)
        else:
            crit = ref.SmoothL1Loss(beta=extra["beta"], reduction=reduction, loss_weight=lw)
        loss = crit(pr, target, weight, avg_factor=avg)
        up = torch.randn(loss.shape, generator=torch.Generator().manual_seed(5)) if loss.dim() else torch.tensor(1.7)
        (loss * up).sum().backward()
        out[f"{name}/pred"] = pred.numpy()
        out[f"{name}/target"] = target.numpy()
        if weight is not None:
            out[f"{name}/weight"] = weight.numpy()
        out[f"{name}/loss"] = loss.detach().numpy()
        out[f"{name}/up"] = up.numpy()
        out[f"{name}/grad"] = pr.grad.numpy()
    np.savez_compressed(os.path.join(HERE, "detloss.npz"), **out)
    print("detloss.npz", len(CASES), "cases", os.path.getsize(os.path.join(HERE, "detloss.npz")), "bytes")


if __name__ == "__main__":
    main()
