"""TEST INFRASTRUCTURE ONLY (CPU oracle) — fp32 restatement of the InvPT forward.

Functional, state-dict-in / tensors-out restatement of

    InvPT/models/transformers/vit.py:172-215, 332-351        plain ViT with cls token, 4 taps
    InvPT/models/transformers/transformer_decoder.py:69-131  scale_embed, preliminary decoder, MLPHead
    InvPT/models/transformers/invpt.py:19-544                UpEmbed, SelfAttention, InvPTBlock/Stage, InvPT
    InvPT/models/transformer_net.py:22-38                    TransformerNet

PINNED like the TaskPrompter oracle: tests/golden/make_golden.py runs the unmodified reference on the `mini`
InvPT config and tests/test_oracle_golden.py checks this file against those fixtures.

Numerics restated exactly: ViT LayerNorm eps 1e-6 (vit.py:254) but nn.LayerNorm default 1e-5 inside InvPT
(invpt.py:256,331,425); attention scale hd^-0.5 in the ViT, dim_out^-0.5 (FULL dim) in InvPT (invpt.py:92);
AvgPool2d(ceil_mode=True); bilinear align_corners=False; BatchNorm eps 1e-5; the dead scale_embed[2] conv,
norm_mt, stage-0 fuse_attn and redu_chan[0] are skipped (their parameters exist but never reach an output).
"""
import torch
import torch.nn.functional as F

from .configs import VIT

BN_EPS = 1e-5


def _ln(x, sd, pre, eps):
    return F.layer_norm(x, (x.shape[-1],), sd[pre + ".weight"], sd[pre + ".bias"], eps)


def _lin(x, sd, pre):
    return F.linear(x, sd[pre + ".weight"], sd.get(pre + ".bias"))


def _bn(x, sd, pre, training, upd):
    w, b = sd[pre + ".weight"], sd[pre + ".bias"]
    if not training:
        return F.batch_norm(x, sd[pre + ".running_mean"], sd[pre + ".running_var"], w, b, False, 0.1, BN_EPS)
    mean = x.mean(dim=(0, 2, 3))
    var = x.var(dim=(0, 2, 3), unbiased=False)
    if upd is not None:
        n = x.numel() // x.shape[1]
        upd[pre + ".running_mean"] = 0.9 * sd[pre + ".running_mean"] + 0.1 * mean.detach()
        upd[pre + ".running_var"] = 0.9 * sd[pre + ".running_var"] + 0.1 * var.detach() * n / max(n - 1, 1)
    return (x - mean[None, :, None, None]) * torch.rsqrt(var[None, :, None, None] + BN_EPS) * w[None, :, None, None] \
        + b[None, :, None, None]


def vit_forward(sd, cfg, img, pre="backbone"):
    """VisionTransformer.forward_features (vit.py:332-351): 4 token maps [B, hw, C] (cls dropped)."""
    C, depth, nH, select = VIT[cfg["backbone"]]
    B = img.shape[0]
    hd = C // nH
    x = F.conv2d(img, sd[pre + ".patch_embed.proj.weight"], sd[pre + ".patch_embed.proj.bias"], stride=16).flatten(2).transpose(1, 2)
    x = torch.cat([sd[pre + ".cls_token"].expand(B, -1, -1), x], dim=1) + sd[pre + ".pos_embed"]
    N = x.shape[1]
    taps = []
    for i in range(depth):
        bp = f"{pre}.blocks.{i}"
        qkv = _lin(_ln(x, sd, bp + ".norm1", 1e-6), sd, bp + ".attn.qkv").view(B, N, 3, nH, hd)
        q, k, v = (qkv[:, :, j].transpose(1, 2) for j in range(3))
        o = torch.softmax(q @ k.transpose(-1, -2) * hd ** -0.5, dim=-1) @ v
        x = x + _lin(o.transpose(1, 2).reshape(B, N, C), sd, bp + ".attn.proj")
        x = x + _lin(F.gelu(_lin(_ln(x, sd, bp + ".norm2", 1e-6), sd, bp + ".mlp.fc1")), sd, bp + ".mlp.fc2")
        if (i + 1) in select:
            taps.append(x[:, 1:])
    taps.append(_ln(x, sd, pre + ".norm", 1e-6)[:, 1:])
    return taps


def _attention(sd, pre, xs, T, heads, kpool, prev, training, upd):
    """SelfAttention.forward (invpt.py:193-241) on per-task NCHW maps xs (already norm1-ed).
    Returns (per-task outputs at the query resolution [B, D, g/2, g/2], attention logits for the next stage)."""
    B, D, g_h, g_w = xs[0].shape
    qs, ks, vs = [], [], []
    for t in range(T):
        cp = f"{pre}.conv_proj_q.{t}"
        qm = _bn(F.conv2d(xs[t], sd[cp + ".conv.weight"], None, stride=2, padding=1, groups=D), sd, cp + ".bn", training, upd)
        qs.append(qm.flatten(2).transpose(1, 2))
        pooled = F.avg_pool2d(xs[t], kpool, kpool, 0, ceil_mode=True).flatten(2).transpose(1, 2)
        ks.append(pooled)
        vs.append(pooled)
    q = _lin(torch.cat(qs, 1), sd, pre + ".proj_q")
    k = _lin(torch.cat(ks, 1), sd, pre + ".proj_k")
    v = _lin(torch.cat(vs, 1), sd, pre + ".proj_v")
    hd = D // heads
    split = lambda t_: t_.view(B, -1, heads, hd).transpose(1, 2)
    q, k, v = split(q), split(k), split(v)
    score = q @ k.transpose(-1, -2) * D ** -0.5                          # scale uses the FULL dim (invpt.py:92)
    if prev is not None:                                                 # message passing (invpt.py:208-229)
        sh, sw = g_h // 4, g_w // 4
        res = sh * sw
        ups = []
        for t in range(T):
            m = prev[:, :, res * t:res * (t + 1), :]                     # [B, heads, sh*sw, keys]
            m = m.permute(0, 1, 3, 2).reshape(B * heads, -1, sh, sw)
            m = F.interpolate(m, scale_factor=2, mode="bilinear", align_corners=False)
            ups.append(m.reshape(B, heads, -1, 4 * res).permute(0, 1, 3, 2))
        score = F.conv2d(torch.cat([score, torch.cat(ups, 2)], 1), sd[pre + ".fuse_attn.weight"], sd[pre + ".fuse_attn.bias"])
    o = torch.softmax(score, dim=-1) @ v
    o = _lin(o.transpose(1, 2).reshape(B, -1, D), sd, pre + ".proj")
    qh, qw = g_h // 2, g_w // 2
    outs = [o[:, qh * qw * t:qh * qw * (t + 1)].transpose(1, 2).reshape(B, D, qh, qw) for t in range(T)]
    return outs, score


def decoder_forward(sd, cfg, taps, pre="multi_task_decoder", training=False, upd=None):
    """TransformerDecoder.forward + InvPT.forward -> ({task: [B, E, 8*mh*... ]}, inter_pred)."""
    C = VIT[cfg["backbone"]][0]
    H, W = cfg["img_size"]
    h, w = H // 16, W // 16
    names = [n for n, _ in cfg["tasks"]]
    T = len(names)
    B = taps[0].shape[0]
    E = cfg["embed_dim"] + cfg["pred_const"]
    to_map = lambda t_: t_.transpose(1, 2).reshape(B, C, h, w)
    back0 = F.conv_transpose2d(to_map(taps[0]), sd[pre + ".scale_embed.0.weight"], sd[pre + ".scale_embed.0.bias"], stride=2, padding=1,
                               output_padding=1)
    back1 = F.conv2d(to_map(taps[1]), sd[pre + ".scale_embed.1.weight"], sd[pre + ".scale_embed.1.bias"], padding=1)
    mh, mw = h // cfg["mtt_down"], w // cfg["mtt_down"]
    x = F.interpolate(to_map(taps[3]), size=(mh, mw), mode="bilinear", align_corners=False)
    feats, inter = {}, {}
    for t in names:
        y = x
        for j in range(2):
            cp = f"{pre}.preliminary_decoder.{t}.{j}"
            y = F.relu(_bn(F.conv2d(y, sd[cp + ".conv.weight"], None, padding=1), sd, cp + ".bn1", training, upd))
        feats[t] = y
        inter[t] = F.conv2d(y, sd[f"{pre}.intermediate_head.{t}.weight"], sd[f"{pre}.intermediate_head.{t}.bias"])
    ip = pre + ".invpt"
    xs = [F.conv2d(torch.cat([feats[t], inter[t]], 1), sd[f"{ip}.mix_proj.{t}.0.weight"], sd[f"{ip}.mix_proj.{t}.0.bias"]) for t in names]
    th, tw = mh * 8, mw * 8
    acc = {t: 0 for t in names}
    prev = None
    dims = [E, E // 2, E // 4]
    for i in range(3):
        sp = f"{ip}.invpt_stages.{i}"
        if i > 0:
            skip = back1 if i == 1 else back0
            for ti in range(T):
                pp = f"{sp}.patch_embed.{ti}.proj"
                y = F.interpolate(xs[ti], scale_factor=2, mode="bilinear", align_corners=False)
                y = F.relu(_bn(F.conv2d(y, sd[pp + ".1.weight"], None, padding=2, dilation=2), sd, pp + ".2", training, upd))
                y = F.relu(_bn(F.conv2d(y, sd[pp + ".4.weight"], None, padding=2, dilation=2), sd, pp + ".5", training, upd))
                xs[ti] = y + skip
        D = dims[i]
        g_h, g_w = xs[0].shape[2:]
        bp = f"{sp}.blocks.0"
        tok = torch.cat([m.flatten(2).transpose(1, 2) for m in xs], 1)                     # [B, T*g*g, D]
        xn = _ln(tok, sd, bp + ".norm1", 1e-5)
        maps = [xn[:, g_h * g_w * t:g_h * g_w * (t + 1)].transpose(1, 2).reshape(B, D, g_h, g_w) for t in range(T)]
        outs, prev = _attention(sd, bp + ".attn", maps, T, 2, 2 ** (i + 1), prev, training, upd)
        up = torch.cat([F.interpolate(o, size=(g_h, g_w), mode="bilinear", align_corners=False).flatten(2).transpose(1, 2) for o in outs], 1)
        tok = tok + up
        tok = tok + _lin(F.gelu(_lin(_ln(tok, sd, bp + ".norm2", 1e-5), sd, bp + ".mlp.fc1")), sd, bp + ".mlp.fc2")
        xs = [tok[:, g_h * g_w * t:g_h * g_w * (t + 1)].transpose(1, 2).reshape(B, D, g_h, g_w) for t in range(T)]
        # LayerNorm over ALL tasks' channels (invpt.py:482,526), then per task redu_chan (i > 0) + resize + sum
        allc = _ln(torch.cat([m.flatten(2).transpose(1, 2) for m in xs], 2), sd, f"{ip}.norm_mts.{i}", 1e-5)
        for ti, t in enumerate(names):
            m = allc[:, :, D * ti:D * (ti + 1)].transpose(1, 2).reshape(B, D, g_h, g_w)
            if i > 0:
                m = F.conv2d(m, sd[f"{ip}.redu_chan.{i}.{ti}.weight"], sd[f"{ip}.redu_chan.{i}.{ti}.bias"])
            acc[t] = acc[t] + F.interpolate(m, size=(th, tw), mode="bilinear", align_corners=False)
    out = {}
    for t in names:
        mp = f"{ip}.mt_proj.{t}"
        out[t] = F.relu(_bn(F.conv2d(acc[t], sd[mp + ".0.weight"], sd[mp + ".0.bias"], padding=1), sd, mp + ".1", training, upd))
    return out, inter


def forward(sd, cfg, img, training=False, bn_updates=None):
    """TransformerNet.forward (transformer_net.py:22-38): {task: [B, n, H, W], 'inter_preds': {...}}."""
    taps = vit_forward(sd, cfg, img)
    feats, inter = decoder_forward(sd, cfg, taps, training=training, upd=bn_updates)
    size = tuple(img.shape[-2:])
    out = {}
    for t, _ in cfg["tasks"]:
        y = F.conv2d(feats[t], sd[f"heads.{t}.linear_pred.weight"], sd[f"heads.{t}.linear_pred.bias"])
        out[t] = F.interpolate(y, size=size, mode="bilinear", align_corners=False)
    out["inter_preds"] = {t: F.interpolate(v, size=size, mode="bilinear", align_corners=False) for t, v in inter.items()}
    return out
