"""TEST INFRASTRUCTURE ONLY (CPU oracle) — fp32 restatement of the TaskPrompter-Swin forward.

Never imported by the product package: only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s
cpu_baseline leg use it, as the checker.  A from-scratch, functional restatement (state-dict in,
tensors out) of the algorithm in

    TaskPrompter/models/transformers/taskprompter_swin.py:120-210   WindowAttention (prompts joined to every window, relative
                                                                     position bias + shift mask on the window x window part only)
    TaskPrompter/models/transformers/taskprompter_swin.py:213-414   SwinTransformerBlock (pad / roll / window partition, raw prompt
                                                                     logits re-assembled as maps, channel attention with softmax)
    TaskPrompter/models/transformers/taskprompter_swin.py:417-472   PatchMerging (features, attention maps, prompts)
    TaskPrompter/models/transformers/taskprompter_swin.py:487-543   BasicLayer
    TaskPrompter/models/transformers/taskprompter_swin.py:546-777   TaskPrompterSwin.forward / cal_task_feature
    TaskPrompter/models/transformers/taskprompter.py:688-715        ConvHead / DEConvHead (shared with the ViT variant)
    TaskPrompter/models/taskprompter_wrapper.py:22-40               TaskPrompterWrapper.forward

PINNED: `tests/golden/make_golden.py` runs the UNMODIFIED reference (imported from /root/reference through
oracle/ref_import.py) on seeded inputs for the miniature configs `mini_swin` / `mini_swin_pad` (oracle/configs.py) and commits
the outputs under tests/golden/; `tests/test_oracle_golden.py` checks this file against them.  The '3ddet' task (FCOS3D head,
mmdet3d) is outside: the reference cannot be run with it here.

Numerics restated exactly: LayerNorm eps 1e-5 (nn.LayerNorm default, :553), GELU = exact erf, BatchNorm eps 1e-5, bilinear
align_corners=False, window attention scale hd^-0.5 applied to the logits BEFORE bias and mask are added (:187-199), mask value
-100 (:299), prompt outputs averaged over the windows (:208), raw logits handed on UNSCALED (:186, :376).
"""
import math

import torch
import torch.nn.functional as F

from . import taskprompter_oracle as tpo

LN_EPS = 1e-5   # taskprompter_swin.py:553 (norm_layer=nn.LayerNorm, default eps)


def _ln(x, sd, pre):
    return F.layer_norm(x, (x.shape[-1],), sd[pre + ".weight"], sd[pre + ".bias"], LN_EPS)


def _lin(x, sd, pre):
    return F.linear(x, sd[pre + ".weight"], sd.get(pre + ".bias"))


def _mlp(sd, pre, t):
    return _lin(F.gelu(_lin(t, sd, pre + ".fc1")), sd, pre + ".fc2")


def relative_position_index(ws):
    """taskprompter_swin.py:147-157: index into the (2 ws - 1)^2 bias table for every (query, key) pair of a ws x ws window."""
    c = torch.stack(torch.meshgrid(torch.arange(ws), torch.arange(ws), indexing="ij")).flatten(1)    # [2, ws*ws]
    rel = (c[:, :, None] - c[:, None, :]).permute(1, 2, 0) + (ws - 1)
    return rel[..., 0] * (2 * ws - 1) + rel[..., 1]


def shift_mask(Hp, Wp, ws, shift):
    """taskprompter_swin.py:281-300: 0 / -100 mask [nW, ws*ws, ws*ws] of the cyclically shifted windows (None without shift)."""
    if shift == 0:
        return None
    img = torch.zeros(Hp, Wp)
    cnt = 0
    for hs in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
        for wsl in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
            img[hs, wsl] = cnt
            cnt += 1
    mw = img.view(Hp // ws, ws, Wp // ws, ws).permute(0, 2, 1, 3).reshape(-1, ws * ws)
    diff = mw[:, None, :] - mw[:, :, None]
    return torch.where(diff != 0, torch.tensor(-100.0), torch.tensor(0.0))


def block_geometry(res, window, shift_flag):
    """taskprompter_swin.py:243-247, 262-272: effective window / shift and the padded resolution of a block."""
    H, W = res
    ws, shift = window, (window // 2 if shift_flag else 0)
    if min(res) <= ws:
        ws, shift = min(res), 0
    Hp, Wp = H + (ws - H % ws) % ws, W + (ws - W % ws) % ws
    return ws, shift, Hp, Wp


def swin_block(sd, pre, x, prompts, res, nH, window, shift_flag, last_block, p, drop=None):
    """taskprompter_swin.py:324-414.  x [B, H*W, C], prompts [B, T, C] -> (x, (raw_spa [B,nH,T,H,W], raw_chan [B,T,C,nh,nw]), prompts)."""
    H, W = res
    B, L, C = x.shape
    T = prompts.shape[1]
    hd = C // nH
    ws, shift, Hp, Wp = block_geometry(res, window, shift_flag)
    spa_p = _ln(prompts, sd, pre + ".norm1")                                  # :331
    chan_p = _lin(prompts, sd, pre + ".token_trans")                          # :332 (raw prompts, not normed)
    xn = _ln(x, sd, pre + ".norm1").view(B, H, W, C)
    xn = F.pad(xn, (0, 0, 0, Wp - W, 0, Hp - H))                              # zeros AFTER the norm (:340-344)
    if shift:
        xn = torch.roll(xn, (-shift, -shift), (1, 2))
    nWh, nWw = Hp // ws, Wp // ws
    win = xn.view(B, nWh, ws, nWw, ws, C).permute(0, 1, 3, 2, 4, 5).reshape(B * nWh * nWw, ws * ws, C)
    nW = nWh * nWw
    seq = torch.cat([spa_p[:, None].expand(B, nW, T, C).reshape(B * nW, T, C), win], 1)     # prompts first (:175-177)
    N = T + ws * ws
    qkv = _lin(seq, sd, pre + ".attn.qkv").view(B * nW, N, 3, nH, hd).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    raw = q @ k.transpose(-2, -1)                                             # unscaled (:186)
    attn = raw * hd ** -0.5
    bias = sd[pre + ".attn.relative_position_bias_table"][relative_position_index(ws).view(-1)].view(ws * ws, ws * ws, nH).permute(2, 0, 1)
    attn = attn.clone()
    attn[:, :, T:, T:] = attn[:, :, T:, T:] + bias[None]
    mask = shift_mask(Hp, Wp, ws, shift)
    if mask is not None:
        a5 = attn.view(B, nW, nH, N, N)
        a5[:, :, :, T:, T:] = a5[:, :, :, T:, T:] + mask[None, :, None]
        attn = a5.view(B * nW, nH, N, N)
    o = (torch.softmax(attn, -1) @ v).transpose(1, 2).reshape(B * nW, N, C)
    o = _lin(o, sd, pre + ".attn.proj")
    new_prompts = o[:, :T].reshape(B, nW, T, C).mean(1)                       # :208
    xo = o[:, T:].reshape(B, nWh, nWw, ws, ws, C).permute(0, 1, 3, 2, 4, 5).reshape(B, Hp, Wp, C)
    raw_spa = raw[:, :, :T, T:].reshape(B, nWh, nWw, nH, T, ws, ws).permute(0, 3, 4, 1, 5, 2, 6).reshape(B, nH, T, Hp, Wp)   # :376
    if shift:
        xo = torch.roll(xo, (shift, shift), (1, 2))
        raw_spa = torch.roll(raw_spa, (shift, shift), (3, 4))
    xo = xo[:, :H, :W].reshape(B, H * W, C)
    raw_spa = raw_spa[:, :, :, :H, :W].contiguous()
    # channel attention (:391-409): keys / values are Linear(pixel_no -> 2 ce) of the attention output's transpose
    ce = p["chan_embed_dim"]
    cq = _lin(chan_p, sd, pre + ".chan_q")                                    # [B, T, ce]
    kv = _lin(xo.transpose(1, 2), sd, pre + ".chan_kv").view(B, C, 2, ce)
    ck, cv = kv[:, :, 0], kv[:, :, 1]
    nh = nw = int(math.isqrt(p["chan_nheads"]))
    r = int(math.isqrt(ce))
    wh, ww = r // nh, r // nw

    def split(t):      # 'b t (nh h nw w) -> b (nh nw) t (h w)'
        return t.view(B, t.shape[1], nh, wh, nw, ww).permute(0, 2, 4, 1, 3, 5).reshape(B, nh * nw, t.shape[1], wh * ww)
    q_, k_, v_ = split(cq), split(ck), split(cv)
    raw_chan = q_ @ k_.transpose(-2, -1)                                      # [B, nh*nw, T, C]
    cx = torch.softmax(raw_chan * ce ** -0.5, -1) @ v_                        # [B, nh*nw, T, wh*ww]
    cx = cx.view(B, nh, nw, T, wh, ww).permute(0, 3, 1, 4, 2, 5).reshape(B, T, ce)
    raw_chan = raw_chan.view(B, nh, nw, T, C).permute(0, 3, 4, 1, 2).contiguous()        # 'b (nh nw) t c -> b t c nh nw'
    # `drop` = optional 4 per-sample scale vectors [B] (already mask / keep) for the block's 4 independent DropPath draws in call order
    # (:412-413, :408-409): x-attention, x-mlp, prompt-attention, prompt-mlp
    dp = (lambda t, i: t) if drop is None else (lambda t, i: t * drop[i][:, None, None])
    x = x + dp(xo, 0)
    x = x + dp(_mlp(sd, pre + ".mlp", _ln(x, sd, pre + ".norm2")), 1)
    if not last_block:
        tp = new_prompts + _lin(_lin(cx, sd, pre + ".chan_proj"), sd, pre + ".token_trans1")
        tp = prompts + dp(tp, 2)
        tp = tp + dp(_mlp(sd, pre + ".mlp", _ln(tp, sd, pre + ".norm2")), 3)
    else:
        tp = new_prompts
    return x, (raw_spa, raw_chan), tp


def patch_merging(sd, pre, x, prompts, attn, res):
    """taskprompter_swin.py:439-472."""
    H, W = res
    B, L, C = x.shape
    xv = x.view(B, H, W, C)
    x = torch.cat([xv[:, 0::2, 0::2], xv[:, 1::2, 0::2], xv[:, 0::2, 1::2], xv[:, 1::2, 1::2]], -1).view(B, -1, 4 * C)
    x = _lin(_ln(x, sd, pre + ".norm"), sd, pre + ".reduction")
    raw_spa, raw_chan = attn
    _, nH, T, _, _ = raw_spa.shape
    raw_spa = F.conv2d(raw_spa.reshape(B, nH * T, H, W), sd[pre + ".spa_attn_ds.weight"], sd[pre + ".spa_attn_ds.bias"], stride=2, padding=1)
    raw_spa = raw_spa.reshape(B, nH, T, H // 2, W // 2)
    raw_chan = _lin(raw_chan.transpose(2, -1), sd, pre + ".process_chan_attn").transpose(2, -1)
    prompts = _lin(prompts, sd, pre + ".task_prompts_up")
    return x, prompts, (raw_spa, raw_chan)


def cal_task_feature(sd, x, attn, il, res, tasks, prompt_len, chans, training, bn_updates):
    """taskprompter_swin.py:715-777 (x [B, h*w, C] -> {task: [B, F, 2h, 2w]})."""
    h, w = res
    B = x.shape[0]
    C = chans
    xm = x.transpose(1, 2).reshape(B, C, h, w)
    spa, chan = attn
    out = {}
    for ti, task in enumerate(tasks):
        a = spa[:, :, ti * prompt_len:(ti + 1) * prompt_len]                 # [B, nH, np, h, w]
        nHp = a.shape[1] * a.shape[2]
        a = a.reshape(B, nHp, h, w)
        gch = C // nHp
        f_spa = xm * (1.0 + a.repeat_interleave(gch, 1))                      # head h scales channels [h*gch, (h+1)*gch)  (:731-735)
        f_spa = F.interpolate(f_spa, scale_factor=2, mode="bilinear", align_corners=False)
        f_spa = F.conv2d(f_spa, sd[f"backbone.fea_decode_spa.{il}.{task}.0.weight"], sd[f"backbone.fea_decode_spa.{il}.{task}.0.bias"])
        ca = chan[:, ti]                                                      # [B, C, nh, nw]
        nh, nw = ca.shape[-2:]
        f_chan = xm * (1.0 + ca.repeat_interleave(h // nh, 2).repeat_interleave(w // nw, 3))
        f_chan = F.interpolate(f_chan, scale_factor=2, mode="bilinear", align_corners=False)
        f_chan = F.conv2d(f_chan, sd[f"backbone.fea_decode_chan.{il}.{task}.0.weight"], sd[f"backbone.fea_decode_chan.{il}.{task}.0.bias"])
        y = torch.cat([f_spa, f_chan], 1)
        pre = f"backbone.fea_fuse.{il}.{task}"
        y = F.conv2d(y, sd[pre + ".0.weight"], sd[pre + ".0.bias"])
        y = F.conv2d(y, sd[pre + ".1.weight"], sd[pre + ".1.bias"], padding=1)
        y = F.gelu(tpo._bn(y, sd, pre + ".2", training, bn_updates))
        y = F.conv2d(y, sd[pre + ".4.weight"], sd[pre + ".4.bias"], padding=1)
        out[task] = y
    return out


def backbone_forward(sd, cfg, img, training=False, bn_updates=None, drop=None):
    """TaskPrompterSwin.forward (:664-713) -> {task: [B, F, H/4', W/4']} at the first level's (x2 upsampled) scale."""
    tasks = [t for t, _ in cfg["tasks"]]
    pl = cfg["prompt_len"]
    ratio = cfg["img_ds_ratio"]
    if ratio != 1:
        img = F.interpolate(img, scale_factor=ratio, mode="bilinear", align_corners=False)
    ps, e = cfg["patch"], cfg["embed"]
    x = F.conv2d(img, sd["backbone.patch_embed.proj.weight"], sd["backbone.patch_embed.proj.bias"], stride=ps)
    B, _, gh, gw = x.shape
    x = _ln(x.flatten(2).transpose(1, 2), sd, "backbone.patch_embed.norm")    # patch_norm=True (:605-607)
    prompts = sd["backbone.task_prompts"][None].expand(B, -1, -1)
    H0, W0 = cfg["img_size"]
    res_out = [[int(H0 // st * ratio), int(W0 // st * ratio)] for st in (8, 16, 32, 32)]   # :594-596 with common_config.py:37-39
    chans = [2 * e, 4 * e, 8 * e, 8 * e]
    pdict = dict(chan_embed_dim=cfg["chan_embed_dim"], chan_nheads=cfg["chan_nheads"])
    fea = {t: [] for t in tasks}
    nl = len(cfg["depths"])
    attn = None
    for il in range(nl):
        res = (gh // 2 ** il, gw // 2 ** il)
        for ib in range(cfg["depths"][il]):
            last = il == nl - 1 and ib == cfg["depths"][il] - 1
            x, attn, prompts = swin_block(sd, f"backbone.layers.{il}.blocks.{ib}", x, prompts, res, cfg["heads"][il], cfg["window"],
                                          ib % 2 == 1, last, pdict, None if drop is None else drop[(il, ib)])
        if il < nl - 1:
            x, prompts, attn = patch_merging(sd, f"backbone.layers.{il}.downsample", x, prompts, attn, res)
            cur = cal_task_feature(sd, x, attn, il, res_out[il], tasks, pl, chans[il], training, bn_updates)
            for t in tasks:
                fea[t].append(cur[t])
    x = _ln(x, sd, "backbone.norm")
    cur = cal_task_feature(sd, x, attn, 3, res_out[3], tasks, pl, chans[3], training, bn_updates)
    out = {}
    for t in tasks:
        fea[t].append(cur[t])
        tgt = fea[t][0].shape[-2:]
        s = sum(F.interpolate(f, tgt, mode="bilinear") for f in fea[t])      # :706 (align_corners default False)
        out[t] = F.conv2d(s, sd[f"backbone.multi_scale_fuse.{t}.weight"], sd[f"backbone.multi_scale_fuse.{t}.bias"], padding=1)
    return out


def forward(sd, cfg, img, training=False, bn_updates=None, target_size=None, drop=None):
    """TaskPrompterWrapper.forward on the Swin backbone -> {task: [B, n_out, H, W]}.  drop: {(layer, block): [4, B] DropPath scales}."""
    fea = backbone_forward(sd, cfg, img, training, bn_updates, drop)
    tgt = target_size or tuple(img.shape[-2:])
    out = {}
    for t, _ in cfg["tasks"]:
        y = tpo.head_forward(sd, f"heads.{t}", cfg["head"], fea[t], training, bn_updates)
        out[t] = F.interpolate(y, tgt, mode="bilinear")
    return out
