"""Host-side mirror of the reference's training criterion (TaskPrompter/losses/loss_functions.py,
loss_schemes.py:9-39, utils/common_config.py:200-236): per-task losses + weighted sum.

`FusedMultiTaskLoss` — the product path (SURVEY.md §8 f rank 1): every task loss is computed straight from the full-resolution
fp32 logits by the HIP kernels mtt_loss_label_stats / mtt_loss_fwd / mtt_loss_bwd (one pass for the loss, one for the gradient,
normalisation constants stay on the device).  Raises on CPU tensors.  (The torch restatement of the reference criterion that the tests
and bench.py's cpu_baseline leg check it against is test infrastructure: oracle/losses_oracle.py.)
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


DEFAULT_WEIGHTS = dict(semseg=1.0, human_parts=2.0, sal=5.0, edge=50.0, normals=10.0, depth=1.0)   # pascal yml:44-50


def _intermediate(p):
    return bool(p.get('intermediate_supervision', False)) if hasattr(p, 'get') else bool(getattr(p, 'intermediate_supervision', False))


def _scheme(loss_of, tasks, all_tasks, weights, pred, gt, intermediate):
    """TaskPrompter/losses/loss_schemes.py:27-39 and InvPT/losses/loss_schemes.py:20-33: weighted sum of the task losses, plus — with
    `p.intermediate_supervision` (InvPT yml:14) — the same losses on the preliminary decoder's `inter_preds` for ALL of the
    criterion's tasks, reported as `inter_<task>` and added to the total with the task weights."""
    out = {t: loss_of(t, pred[t], gt[t]) for t in tasks}
    out['total'] = torch.sum(torch.stack([weights[t] * out[t] for t in tasks]))
    if intermediate:
        inter = pred['inter_preds']
        for t in all_tasks:
            v = loss_of(t, inter[t], gt[t])
            out['inter_%s' % t] = v
            out['total'] = out['total'] + weights[t] * v
    return out


_KIND = dict(ce=0, ce_balanced=1, bce=2, l1=3, l1_norm=4)


class _TaskLossFn(torch.autograd.Function):
    """One task's loss on the HIP kernels; backward = mtt_loss_bwd scaled by the incoming scalar gradient (read on the device)."""

    @staticmethod
    def forward(ctx, pred, label, kind, ignore, pos_weight):
        from . import ops
        pred = pred.contiguous()
        label = label.contiguous().float()
        if pred.dtype != torch.float32:
            raise RuntimeError("fused losses take the fp32 logits the HIP path produces")
        B, C = pred.shape[0], pred.shape[1]
        HW = pred[0, 0].numel()
        Cl = label.shape[1]
        stats = torch.zeros(2, dtype=torch.float32, device=pred.device)
        loss = torch.zeros(1, dtype=torch.float32, device=pred.device)
        kw = dict(pred=pred, label=label, dpred=None, loss=loss, stats=stats, B=B, HW=HW, C=C, Cl=Cl, kind=kind, ignore=float(ignore),
                  pos_weight=float(pos_weight), ws=ops.ws_for("loss", pred.device, B=B, HW=HW))
        ops.call("loss_label_stats", xargs=[stats], **kw)
        ops.call("loss_fwd", **kw)
        ctx.save_for_backward(pred, label, stats)
        ctx.meta = (kind, float(ignore), float(pos_weight))
        return loss[0]

    @staticmethod
    def backward(ctx, gout):
        from . import ops
        pred, label, stats = ctx.saved_tensors
        kind, ignore, pos_weight = ctx.meta
        dpred = torch.empty_like(pred)
        g = gout.reshape(1).float().contiguous()
        ops.call("loss_bwd", pred=pred, label=label, dpred=dpred, loss=None, stats=stats, B=pred.shape[0], HW=pred[0, 0].numel(),
                 C=pred.shape[1], Cl=label.shape[1], kind=kind, ignore=ignore, pos_weight=pos_weight, xargs=[g])
        return dpred, None, None, None, None


def _fused_spec(p, task):
    ign = p.get('ignore_index', 255)
    if task == 'edge':
        return _KIND['bce'], ign, p.get('edge_w', 0.95)
    if task in ('semseg', 'human_parts'):
        return _KIND['ce'], ign, 0.0
    if task == 'normals':
        return _KIND['l1_norm'], ign, 0.0
    if task == 'sal':
        return _KIND['ce_balanced'], ign, 0.0
    if task == 'depth':
        return _KIND['l1'], -1, 0.0
    raise NotImplementedError(task)


class FusedMultiTaskLoss(nn.Module):
    """loss_schemes.py:9-39 with the per-task losses of utils/common_config.py:200-228 on the HIP kernels (same call signature and
    result dict as the reference's MultiTaskLoss)."""

    def __init__(self, p, tasks, loss_weights=None):
        super().__init__()
        self.tasks = list(tasks)
        self.spec = {t: _fused_spec(p, t) for t in self.tasks}
        self.loss_weights = dict(loss_weights or {t: DEFAULT_WEIGHTS[t] for t in self.tasks})
        self.intermediate_supervision = _intermediate(p)

    def forward(self, pred, gt, tasks=None):
        return _scheme(lambda t, a, b: _TaskLossFn.apply(a, b, *self.spec[t]), tasks or self.tasks, self.tasks, self.loss_weights,
                       pred, gt, self.intermediate_supervision)


def synthetic_targets(p, B, H, W, device, seed=0):
    """Synthetic labels of the shapes / value conventions the reference's datasets produce."""
    g = torch.Generator(device='cpu').manual_seed(seed)
    gt = {}
    for t in p.TASKS.NAMES:
        n = p.TASKS.NUM_OUTPUT[t]
        ign = torch.rand(B, 1, H, W, generator=g) < 0.05
        if t in ('semseg', 'human_parts'):
            y = torch.randint(0, n, (B, 1, H, W), generator=g).float()
            y[ign] = 255
        elif t == 'sal':
            y = torch.randint(0, 2, (B, 1, H, W), generator=g).float()
        elif t == 'edge':
            y = (torch.rand(B, 1, H, W, generator=g) < 0.05).float()
        elif t == 'normals':
            y = F.normalize(torch.randn(B, 3, H, W, generator=g), dim=1)
            y[ign.expand(B, 3, H, W)] = 255
        elif t == 'depth':
            y = torch.rand(B, 1, H, W, generator=g) * 9.9 + 0.1
            y[ign] = -1
        else:
            raise NotImplementedError(t)
        gt[t] = y.to(device)
    return gt
