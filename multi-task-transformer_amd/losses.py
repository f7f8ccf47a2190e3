"""Host-side mirror of the reference's training criterion (TaskPrompter/losses/loss_functions.py,
loss_schemes.py:9-39, utils/common_config.py:200-236): per-task losses + weighted sum.

Row (f) rank 1 of SURVEY.md §8 ("next": fused loss-from-logits kernels).  In round 1 these are restated with
torch ops on the full-resolution fp32 logits the HIP path produces; they are part of the timed training step.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


class CrossEntropyLoss(nn.Module):
    """loss_functions.py:15-55 — CE with ignore index; `balanced` = 2-class re-weighting by label frequency (sal)."""

    def __init__(self, ignore_index=255, balanced=False):
        super().__init__()
        self.ignore_index, self.balanced = ignore_index, balanced

    def forward(self, out, label):
        label = label.squeeze(1).long()
        valid = label != self.ignore_index
        weight = None
        if self.balanced:
            lv = label[valid].float()
            w_pos = (1.0 - lv).sum() / max(lv.numel(), 1)
            weight = torch.stack((1.0 - w_pos, w_pos))
        loss = F.cross_entropy(out, label, weight=weight, ignore_index=self.ignore_index, reduction='none')
        return loss.sum() / valid.sum().clamp_min(1)


class BalancedBinaryCrossEntropyLoss(nn.Module):
    """loss_functions.py:57-89 (edge): HED-style balanced BCE with a fixed positive weight."""

    def __init__(self, pos_weight=0.95, ignore_index=255):
        super().__init__()
        self.pos_weight, self.ignore_index = pos_weight, ignore_index

    def forward(self, output, label):
        mask = label != self.ignore_index
        w = torch.as_tensor(self.pos_weight, device=output.device, dtype=output.dtype)
        factor = 1.0 / (1.0 - w)
        loss = F.binary_cross_entropy_with_logits(output[mask], label[mask], pos_weight=w * factor, reduction='mean')
        return loss / factor


class L1Loss(nn.Module):
    """loss_functions.py:143-177 (normals: normalize=True, ignore 255; depth: ignore -1)."""

    def __init__(self, normalize=False, ignore_index=255):
        super().__init__()
        self.normalize, self.ignore_index = normalize, ignore_index

    def forward(self, out, label):
        if self.normalize:
            out = F.normalize(out, p=2, dim=1)
        mask = (label != self.ignore_index).all(dim=1, keepdim=True).expand_as(out)
        n_valid = (label != self.ignore_index).all(dim=1).sum().clamp_min(1)
        return (out - label).abs()[mask].sum() / n_valid


def get_loss(p, task):
    """TaskPrompter/utils/common_config.py:200-228."""
    ign = p.get('ignore_index', 255)
    if task == 'edge':
        return BalancedBinaryCrossEntropyLoss(pos_weight=p.get('edge_w', 0.95), ignore_index=ign)
    if task in ('semseg', 'human_parts'):
        return CrossEntropyLoss(ignore_index=ign)
    if task == 'normals':
        return L1Loss(normalize=True, ignore_index=ign)
    if task == 'sal':
        return CrossEntropyLoss(balanced=True, ignore_index=ign)
    if task == 'depth':
        return L1Loss(ignore_index=-1)
    raise NotImplementedError(task)


DEFAULT_WEIGHTS = dict(semseg=1.0, human_parts=2.0, sal=5.0, edge=50.0, normals=10.0, depth=1.0)   # pascal yml:44-50


class MultiTaskLoss(nn.Module):
    """loss_schemes.py:9-39 (dense tasks)."""

    def __init__(self, p, tasks, loss_weights=None):
        super().__init__()
        self.tasks = list(tasks)
        self.loss_ft = nn.ModuleDict({t: get_loss(p, t) for t in self.tasks})
        self.loss_weights = dict(loss_weights or {t: DEFAULT_WEIGHTS[t] for t in self.tasks})

    def forward(self, pred, gt, tasks=None):
        tasks = tasks or self.tasks
        out = {t: self.loss_ft[t](pred[t], gt[t]) for t in tasks}
        out['total'] = torch.sum(torch.stack([self.loss_weights[t] * out[t] for t in tasks]))
        return out


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
