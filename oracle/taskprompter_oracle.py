"""TEST INFRASTRUCTURE ONLY (CPU oracle) — fp32 restatement of the TaskPrompter forward.

Never imported by the product package: only `tests/`, `__graft_entry__.smoke()` and
`bench.py`'s cpu_baseline leg use it, as the checker.  It is a from-scratch, functional,
token-major restatement (state-dict in, tensors out) of the algorithm in

    TaskPrompter/models/transformers/taskprompter.py:168-254   Attention
    TaskPrompter/models/transformers/taskprompter.py:257-279   Block
    TaskPrompter/models/transformers/taskprompter.py:392-487   TaskPrompter.forward / cal_task_feature
    TaskPrompter/models/transformers/taskprompter.py:688-715   ConvHead / DEConvHead
    TaskPrompter/models/taskprompter_wrapper.py:22-40          TaskPrompterWrapper.forward

PINNED: `tests/golden/make_golden.py` runs the UNMODIFIED reference (imported from
/root/reference through oracle/ref_import.py) on seeded inputs and commits the input/output
tensors under tests/golden/; `tests/test_oracle_golden.py` checks this file against them.
The reference itself ships no golden vectors or known-answer tests.

Numerics restated exactly: LayerNorm eps 1e-6, GELU = exact erf, BatchNorm eps 1e-5 /
momentum 0.1, bilinear align_corners=False, attention scale hd^-0.5 on the softmax only
(the logits handed to the decoder are the UNSCALED q.k), prompts first in the sequence.
"""
import math

import torch
import torch.nn.functional as F

from .configs import VIT

LN_EPS = 1e-6   # taskprompter.py:310
BN_EPS = 1e-5


def _ln(x, sd, pre):
    return F.layer_norm(x, (x.shape[-1],), sd[pre + ".weight"], sd[pre + ".bias"], LN_EPS)


def _lin(x, sd, pre):
    return F.linear(x, sd[pre + ".weight"], sd.get(pre + ".bias"))


def _bn(x, sd, pre, training, bn_updates):
    """BatchNorm2d on NCHW.  training=True uses biased batch stats (and records the running-stat
    update the module would make, momentum 0.1 / unbiased variance) exactly like nn.BatchNorm2d."""
    w, b = sd[pre + ".weight"], sd[pre + ".bias"]
    if not training:
        return F.batch_norm(x, sd[pre + ".running_mean"], sd[pre + ".running_var"], w, b, False, 0.1, BN_EPS)
    mean = x.mean(dim=(0, 2, 3))
    var = x.var(dim=(0, 2, 3), unbiased=False)
    if bn_updates is not None:
        n = x.numel() // x.shape[1]
        bn_updates[pre + ".running_mean"] = 0.9 * sd[pre + ".running_mean"] + 0.1 * mean.detach()
        bn_updates[pre + ".running_var"] = 0.9 * sd[pre + ".running_var"] + 0.1 * var.detach() * n / max(n - 1, 1)
    xh = (x - mean[None, :, None, None]) * torch.rsqrt(var[None, :, None, None] + BN_EPS)
    return xh * w[None, :, None, None] + b[None, :, None, None]


def attention(sd, pre, xn, pn, num_heads, grid, chan_nheads):
    """taskprompter.py:195-254.  xn [B,hw,C], pn [B,T,C] are ALREADY norm1-ed (Block :272).

    Returns (x_out [B,hw,C], prompt_out [B,T,C], raw_spa [B,nH,T,N] = unscaled q.k of the prompt
    rows — the only rows cal_task_feature reads — and raw_chan [B,T,C,nh,nw])."""
    B, hw, C = xn.shape
    T = pn.shape[1]
    hd = C // num_heads
    seq = torch.cat([pn, xn], dim=1)                                   # prompts FIRST (:199)
    N = T + hw
    qkv = _lin(seq, sd, pre + ".qkv").view(B, N, 3, num_heads, hd)
    q, k, v = (qkv[:, :, i].transpose(1, 2) for i in range(3))       # [B,nH,N,hd]
    raw = q @ k.transpose(-1, -2)                                     # unscaled (:204)
    o = torch.softmax(raw * hd ** -0.5, dim=-1) @ v
    o = _lin(o.transpose(1, 2).reshape(B, N, C), sd, pre + ".proj")
    p_out, x_out = o[:, :T], o[:, T:]
    # channel attention (:216-250): queries = token_trans(norm1(prompts)), keys = norm1(x)^T, windowed
    h, w = grid
    nh = nw = int(math.isqrt(chan_nheads))
    wh, ww = h // nh, w // nw
    cq = _lin(pn, sd, pre + ".token_trans")                           # [B,T,hw]
    cq_w = cq.view(B, T, nh, wh, nw, ww)
    xk_w = xn.view(B, nh, wh, nw, ww, C)
    raw_chan = torch.einsum("btiajk,biajkc->btcij", cq_w, xk_w)       # [B,T,C,nh,nw], unscaled (:240,246)
    p_out = p_out + _lin(cq, sd, pre + ".token_trans1")               # (:250)
    return x_out, p_out, raw[:, :, :T, :], raw_chan


def block(sd, pre, x, p, num_heads, grid, chan_nheads, drop=None):
    """taskprompter.py:270-279.  `drop` = optional 4 per-sample scale vectors [B] (already mask/keep)
    for the reference's 4 independent DropPath draws in call order: x-attn, x-mlp, prompt-attn, prompt-mlp."""
    xa, pa, raw_spa, raw_chan = attention(sd, pre + ".attn", _ln(x, sd, pre + ".norm1"), _ln(p, sd, pre + ".norm1"),
                                          num_heads, grid, chan_nheads)

    def mlp(t):
        return _lin(F.gelu(_lin(_ln(t, sd, pre + ".norm2"), sd, pre + ".mlp.fc1")), sd, pre + ".mlp.fc2")

    def dp(t, i):
        return t if drop is None else t * drop[i][:, None, None]

    x = x + dp(xa, 0)
    x = x + dp(mlp(x), 1)
    p = p + dp(pa, 2)
    p = p + dp(mlp(p), 3)
    return x, p, raw_spa, raw_chan


def task_features(sd, pre, cfg, x, raw_spa, raw_chan, il, training, bn_updates):
    """cal_task_feature (taskprompter.py:424-487) for tap `il`.  x [B,hw,C] -> {task: [B,F,h,w]} (NCHW)."""
    C, _, nH, _ = VIT[cfg["backbone"]]
    H, W = cfg["img_size"]
    h, w = H // 16, W // 16
    B = x.shape[0]
    names = [n for n, _ in cfg["tasks"]]
    T = len(names)
    hd = C // nH
    nh = nw = int(math.isqrt(cfg["chan_nheads"]))
    feats = []
    for t, task in enumerate(names):
        # spatial modulation (:436-446): channel c of head c//hd is scaled by (1 + logit of prompt t vs pixel)
        a = raw_spa[:, :, t, T:]                                      # [B,nH,hw]
        f_spa = x * (1.0 + a.transpose(1, 2).repeat_interleave(hd, dim=2))
        # channel modulation (:452-467): channel c in window (i,j) scaled by (1 + chan logit[t,c,i,j])
        bwin = raw_chan[:, t]                                         # [B,C,nh,nw]
        bmap = bwin.repeat_interleave(h // nh, dim=2).repeat_interleave(w // nw, dim=3)   # [B,C,h,w]
        f_chan = x * (1.0 + bmap.flatten(2).transpose(1, 2))
        k = f"{pre}.fea_decode_spa.{il}.{task}.0"
        d_spa = F.linear(f_spa, sd[k + ".weight"].flatten(1), sd[k + ".bias"])
        k = f"{pre}.fea_decode_chan.{il}.{task}.0"
        d_chan = F.linear(f_chan, sd[k + ".weight"].flatten(1), sd[k + ".bias"])
        k = f"{pre}.fea_fuse.{il}.{task}"
        y = F.linear(torch.cat([d_spa, d_chan], dim=2), sd[k + ".0.weight"].flatten(1), sd[k + ".0.bias"])
        y = y.transpose(1, 2).reshape(B, -1, h, w)
        y = F.conv2d(y, sd[k + ".1.weight"], sd[k + ".1.bias"], padding=1)
        y = F.gelu(_bn(y, sd, k + ".2", training, bn_updates))
        y = F.conv2d(y, sd[k + ".4.weight"], sd[k + ".4.bias"])
        feats.append(y)
    if cfg["use_ctr"]:
        # cross-task reweighting (:478-485): prompt<->prompt raw logits -> per-head MLP -> TxT mixing
        mixed = []
        for t, task in enumerate(names):
            k = f"{pre}.ctr_attn_conv.{il}.{task}"
            z = raw_spa[:, :, t, :T].transpose(1, 2)                  # [B,T(target),nH]
            z = F.gelu(F.linear(z, sd[k + ".0.weight"].flatten(1), sd[k + ".0.bias"]))
            wgt = F.linear(z, sd[k + ".2.weight"].flatten(1), sd[k + ".2.bias"])[..., 0]   # [B,T]
            mixed.append(sum(wgt[:, s, None, None, None] * feats[s] for s in range(T)))
        feats = mixed
    return dict(zip(names, feats))


def backbone_forward(sd, cfg, img, pre="backbone", training=False, drop=None, bn_updates=None):
    """TaskPrompter.forward (taskprompter.py:392-422): -> {task: [B,F,4h,4w]}."""
    C, depth, nH, select = VIT[cfg["backbone"]]
    H, W = cfg["img_size"]
    grid = (H // 16, W // 16)
    B = img.shape[0]
    x = F.conv2d(img, sd[pre + ".patch_embed.proj.weight"], sd[pre + ".patch_embed.proj.bias"], stride=16)
    x = x.flatten(2).transpose(1, 2) + sd[pre + ".pos_embed"][:, 1:]   # cls slot unused (:394)
    p = sd[pre + ".task_prompts"][None].expand(B, -1, -1)
    acc = None
    raw_spa = raw_chan = None
    for i in range(depth):
        x, p, raw_spa, raw_chan = block(sd, f"{pre}.blocks.{i}", x, p, nH, grid, cfg["chan_nheads"],
                                        None if drop is None else drop[i])
        if (i + 1) in select:
            il = list(select).index(i + 1)
            cur = task_features(sd, pre, cfg, x, raw_spa, raw_chan, il, training, bn_updates)
            acc = cur if acc is None else {t: acc[t] + cur[t] for t in cur}
    xf = _ln(x, sd, pre + ".norm")
    cur = task_features(sd, pre, cfg, xf, raw_spa, raw_chan, 3, training, bn_updates)   # last block's attention (:417)
    return {t: F.interpolate(acc[t] + cur[t], scale_factor=4, mode="bilinear", align_corners=False) for t in cur}


def head_forward(sd, pre, kind, f, training=False, bn_updates=None):
    """ConvHead (taskprompter.py:688-698) / DEConvHead (:700-715) on NCHW features."""
    if kind == "conv":
        y = F.conv2d(f, sd[pre + ".mt_proj.0.weight"], sd[pre + ".mt_proj.0.bias"], padding=1)
        y = F.gelu(_bn(y, sd, pre + ".mt_proj.1", training, bn_updates))
    else:
        y = F.conv_transpose2d(f, sd[pre + ".mt_proj.0.weight"], sd[pre + ".mt_proj.0.bias"], stride=2)
        y = F.gelu(_bn(y, sd, pre + ".mt_proj.1", training, bn_updates))
        y = F.conv2d(y, sd[pre + ".mt_proj.3.weight"], sd[pre + ".mt_proj.3.bias"], padding=1)
        y = F.gelu(_bn(y, sd, pre + ".mt_proj.4", training, bn_updates))
    return F.conv2d(y, sd[pre + ".linear_pred.weight"], sd[pre + ".linear_pred.bias"])


def forward(sd, cfg, img, training=False, drop=None, bn_updates=None, return_features=False):
    """TaskPrompterWrapper.forward (taskprompter_wrapper.py:22-40): {task: [B,n_out,H,W]} fp32; (H, W) = the input size, or
    cfg["dd_label_map_size"] when the config carries one (taskprompter_wrapper.py:17-20, :26)."""
    feats = backbone_forward(sd, cfg, img, "backbone", training, drop, bn_updates)
    out = {}
    target = tuple(cfg["dd_label_map_size"]) if cfg.get("dd_label_map_size") is not None else tuple(img.shape[-2:])
    for task, _ in cfg["tasks"]:
        y = head_forward(sd, f"heads.{task}", cfg["head"], feats[task], training, bn_updates)
        out[task] = F.interpolate(y, size=target, mode="bilinear", align_corners=False)
    return (out, feats) if return_features else out
