"""Training-mode BatchNorm2d over a stack of NHWC maps [Z, rows, ld] — one nn.BatchNorm2d / nn.SyncBatchNorm holder per map
(taskprompter.py:362,692,705; InvPT's SyncBatchNorm, invpt.py:14) — shared by the no-grad and the autograd paths.

Statistics are centred (mean, M2 = sum (x - mean)^2) from the deterministic two-level reduction mtt_bn_stats, like the
reference's nn.BatchNorm2d (never E[x^2] - E[x]^2).  Under SyncBatchNorm + an initialised process group ALL Z maps of a stage
exchange their (mean, M2, count) triplets in ONE collective (forward: an all_reduce of a rank-slotted table) and their backward sums in ONE all_reduce, and the
per-rank triplets are merged exactly (Chan), so ranks may hold different numbers of rows.
"""
import torch
import torch.distributed as dist
import torch.nn as nn

from . import ops


def _group(bns):
    """process group of a BatchNorm stage: nn.SyncBatchNorm.process_group (convert_sync_batchnorm(model, process_group=...)), default WORLD.
    One collective serves the whole stage, so its holders must agree."""
    gs = {id(getattr(bn, "process_group", None)): getattr(bn, "process_group", None) for bn in bns}
    if len(gs) > 1:
        raise ValueError("the SyncBatchNorm holders of one stage (one per task) must share a process group")
    return next(iter(gs.values()))


def _world(bns):
    bn = bns[0]
    sync = isinstance(bn, nn.SyncBatchNorm) or getattr(bn, "_mtt_sync", False)     # _mtt_sync: CPU/gloo tests (DDP rejects SyncBN on CPU)
    if sync and dist.is_available() and dist.is_initialized():
        w = dist.get_world_size(_group(bns))
        return w if w > 1 else 1
    return 1


def train_stats(x, C, bns):
    """x [Z, rows, ld] -> (mean [Z, C], rstd [Z, C], scale) and the running-stat update of every holder.
    `scale` = local rows / global rows ([Z, 1] tensor, or 1.0): factor for the backward sums, whose kernel divides by the local
    row count."""
    Z, rows, _ = x.shape
    mean, m2 = ops.bn_stats(x, C)
    world = _world(bns)
    if world > 1:
        # one collective for the whole stage: every rank fills its own slot of a zero [W, Z, 2C+1] table and the table is all-reduced
        # (sum with zeros is exact).  An all_reduce rather than an all_gather because it is the one collective every backend offers
        # on device tensors (gloo has no device all_gather; RCCL has both) — the table is a few KB.
        allp = torch.zeros(world, Z, 2 * C + 1, dtype=torch.float32, device=x.device)
        grp = _group(bns)
        allp[dist.get_rank(grp)] = torch.cat([mean, m2, torch.full((Z, 1), float(rows), dtype=torch.float32, device=x.device)], 1)
        dist.all_reduce(allp, group=grp)
        cnt = allp[:, :, 2 * C:]
        n = cnt.sum(0)                                                 # [Z, 1]
        mean = (allp[:, :, :C] * cnt).sum(0) / n
        m2 = (allp[:, :, C:2 * C] + cnt * (allp[:, :, :C] - mean) ** 2).sum(0)
        scale = float(rows) / n
        unbias = 1.0 / (n - 1.0).clamp_min(1.0)
        inv_n = 1.0 / n
    else:
        scale = 1.0
        unbias = 1.0 / max(rows - 1, 1)
        inv_n = 1.0 / rows
    var = m2 * inv_n
    eps = torch.tensor([bn.eps for bn in bns], dtype=torch.float32, device=x.device)[:, None] if len({bn.eps for bn in bns}) > 1 else bns[0].eps
    rstd = torch.rsqrt(var + eps)
    with torch.no_grad():
        tracked = [bn for bn in bns if bn.track_running_stats and bn.running_mean is not None]
        if tracked:
            torch._foreach_add_([bn.num_batches_tracked for bn in tracked], 1)
            var_u = m2 * unbias
            groups = {}                                    # momentum -> (running buffers, batch statistics): one foreach pair per group
            for z, bn in enumerate(bns):
                if bn not in tracked:
                    continue
                m = bn.momentum if bn.momentum is not None else 1.0 / float(bn.num_batches_tracked)   # None: cumulative average
                run, new = groups.setdefault(m, ([], []))
                run += [bn.running_mean, bn.running_var]
                new += [mean[z], var_u[z]]
            for m, (run, new) in groups.items():           # running = (1 - m) * running + m * batch  (nn.BatchNorm2d)
                torch._foreach_mul_(run, 1 - m)
                torch._foreach_add_(run, new, alpha=m)
    return mean.contiguous(), rstd.contiguous(), scale


def train_forward(x, C, bns, act, gammas=None, betas=None):
    """y = act(BN_train(x)) for the stack; returns (y, mean, rstd, scale)."""
    mean, rstd, scale = train_stats(x, C, bns)
    if gammas is None:
        gammas = ops.stack_vec([bn.weight for bn in bns], ('bng',))
        betas = ops.stack_vec([bn.bias for bn in bns], ('bnb',))
    y = ops.bn_apply(x, C, mean, rstd, gammas, betas, act)
    return y, mean, rstd, scale


def sync_backward_sums(s, bns):
    """s [2, Z, C] local sums of du and du*xhat -> summed over ranks (one all_reduce per stage)."""
    if _world(bns) > 1:
        dist.all_reduce(s, group=_group(bns))
    return s
