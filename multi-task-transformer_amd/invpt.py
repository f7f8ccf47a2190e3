"""MI355X-native InvPT: same nn.Module API / state_dict layout as the reference
(InvPT/models/transformers/vit.py, transformer_decoder.py, invpt.py, InvPT/models/transformer_net.py),
executed on the libmtt_hip.so kernels.

Schedule (differs from the reference's op graph):
  * ViT: one fp32 token buffer [B, 1+hw, C] (cls first), LayerNorm -> qkv GEMM -> flash attention -> proj GEMM
    (+residual) -> LayerNorm -> fc1+GELU -> fc2 (+residual); taps are copies of the patch rows.
  * Decoder feature maps are task-major NHWC stacks [T, B*g*g, pitch(D)]; every per-task conv / BN / 1x1 is one
    task-batched launch; the shared-weight Linears (proj_q/k/v, proj, MLP) run once over all tasks' tokens.
  * Cross-task attention (heads = 2, head dim D/2 not 64) uses the batched GEMM + row-softmax kernels on
    batch-major q/k/v ([B, T*q, .]) that the projection GEMMs produce through their row-group output mapping;
    heads are padded to 8-column multiples.  The message passing of invpt.py:208-229 is one fused kernel.
  * Dead reference compute is skipped (scale_embed[2], norm_mt, stage-0 fuse_attn, redu_chan[0]); their parameters
    exist (strict state_dict) and — as in the reference — receive no gradient.

This file is the no-grad forward (eval and train-mode BatchNorm statistics); with gradients enabled the same
modules route to invpt_autograd.py (autograd Functions with hand-written backward on the same kernels).
"""
import math
from collections import OrderedDict
from functools import partial

import torch
import torch.nn as nn

from . import ops
from ._lib import ACT_GELU, ACT_NONE, ACT_RELU, F32, OP_K, OP_R, dtype_code
from .taskprompter import Mlp, PatchEmbed, _init_vit_weights, _prec_of, trunc_normal_

BATCHNORM = nn.SyncBatchNorm      # invpt.py:14, transformer_decoder.py:13 (parameter holder; statistics are computed by the kernels)
pitch = ops.pitch


# ---------------------------------------------------------------------------------------------------------
# ViT backbone (vit.py:172-351)
# ---------------------------------------------------------------------------------------------------------
class Attention(nn.Module):
    def __init__(self, dim, num_heads=8, qkv_bias=False):
        super().__init__()
        self.num_heads = num_heads
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim)


class Block(nn.Module):
    def __init__(self, dim, num_heads, mlp_ratio=4., qkv_bias=False, drop_path=0., norm_layer=nn.LayerNorm):
        super().__init__()
        self.norm1 = norm_layer(dim)
        self.attn = Attention(dim, num_heads=num_heads, qkv_bias=qkv_bias)
        self.drop_path_rate = float(drop_path)
        self.norm2 = norm_layer(dim)
        self.mlp = Mlp(dim, int(dim * mlp_ratio))


class VisionTransformer(nn.Module):
    """vit.py:218-351; forward(x) -> (x_last [B, hw, C], [4 taps [B, hw, C]])."""

    def __init__(self, select_list, img_size=224, patch_size=16, in_chans=3, num_classes=1000, embed_dim=768, depth=12,
                 num_heads=12, mlp_ratio=4., qkv_bias=True, representation_size=None, distilled=False, drop_rate=0.,
                 attn_drop_rate=0., drop_path_rate=0., embed_layer=None, norm_layer=None, act_layer=None, weight_init='',
                 prec=ops.DEFAULT_PREC):
        super().__init__()
        assert patch_size == 16 and embed_dim // num_heads == 64 and not distilled
        self.num_features = self.embed_dim = embed_dim
        self.num_heads = num_heads
        norm_layer = norm_layer or partial(nn.LayerNorm, eps=1e-6)
        if isinstance(img_size, int):
            img_size = (img_size, img_size)
        self.patch_embed = PatchEmbed(img_size=img_size, patch_size=patch_size, in_chans=in_chans, embed_dim=embed_dim)
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.pos_embed = nn.Parameter(torch.zeros(1, self.patch_embed.num_patches + 1, embed_dim))
        dpr = [x.item() for x in torch.linspace(0, drop_path_rate, depth)]
        self.blocks = nn.Sequential(*[Block(embed_dim, num_heads, mlp_ratio, qkv_bias, dpr[i], norm_layer) for i in range(depth)])
        self.norm = norm_layer(embed_dim)
        self.select_list = list(select_list)
        trunc_normal_(self.pos_embed, std=.02)
        trunc_normal_(self.cls_token, std=.02)
        self.apply(_init_vit_weights)
        self.prec = ops.Prec(prec)

    def load_pretrained(self, checkpoint_path, prefix=''):
        """vit.py:320-321: import a Google/Flax ViT .npz (class token, resized position embedding)."""
        from .checkpoints import load_flax_vit_npz
        return load_flax_vit_npz(self, checkpoint_path, prefix)

    def forward(self, img):
        taps = self.forward_taps(img)
        B = img.shape[0]
        return taps[-1].view(B, -1, self.embed_dim).float(), [t.view(B, -1, self.embed_dim).float() for t in taps]

    def forward_taps(self, img):
        """-> 4 contiguous [B*hw, C] activation-dtype token maps (cls dropped), vit.py:340-349."""
        if torch.is_grad_enabled() and (img.requires_grad or any(q.requires_grad for q in self.parameters())):
            from . import invpt_autograd
            return invpt_autograd.vit_taps(self, img)
        prec = self.prec
        B = img.shape[0]
        C, nH = self.embed_dim, self.num_heads
        hw = self.patch_embed.num_patches
        N = hw + 1
        XT = torch.empty(B * N, C, dtype=torch.float32, device=img.device)
        XT.view(B, N, C)[:, :1] = (self.cls_token + self.pos_embed[:, :1]).detach()
        cols = ops.patchify(img.float(), prec)
        ops.linear(cols, ops.pack_linear([self.patch_embed.proj.weight], prec, 'vpe'), C, prec,
                   bias=self.patch_embed.proj.bias.detach()[None], out=XT.view(B, N, C)[:, 1:], d_rows=(hw, N * C, C),
                   resid=self.pos_embed.detach()[0, 1:], r_rows=(hw, 0, C), M=B * hw)
        taps = []
        # x3f: the four big Linears of a block on the split-plane LDS-DMA kernel (operands written as hi / lo planes by LayerNorm, the
        # qkv / fc1 epilogues and the attention kernel), as TaskPrompter._block_split; the K = C reductions must be whole 32-deep steps
        split = prec.split and C % 32 == 0
        pack = (lambda ws, tg: ops.pack_linear_split(ws, tg)) if split else (lambda ws, tg: ops.pack_linear(ws, prec, tg))
        sp = dict(out_dtype="split") if split else {}
        for i, blk in enumerate(self.blocks):
            tag = ('vblk', i)
            xn, _, _ = ops.layernorm(XT, blk.norm1.weight.detach(), blk.norm1.bias.detach(), blk.norm1.eps, prec, **sp)
            qkv = ops.linear(xn, pack([blk.attn.qkv.weight], tag + ('qkv',)), 3 * C, prec, bias=blk.attn.qkv.bias.detach()[None], **sp)[0]
            ao, _, _ = ops.attention(qkv, B, N, nH, 0, prec)
            XT2 = torch.empty_like(XT)
            ops.linear(ao, pack([blk.attn.proj.weight], tag + ('proj',)), C, prec, bias=blk.attn.proj.bias.detach()[None], out=XT2, resid=XT)
            xn2, _, _ = ops.layernorm(XT2, blk.norm2.weight.detach(), blk.norm2.bias.detach(), blk.norm2.eps, prec, **sp)
            hmid = ops.linear(xn2, pack([blk.mlp.fc1.weight], tag + ('fc1',)), 4 * C, prec, bias=blk.mlp.fc1.bias.detach()[None],
                              act=ACT_GELU, **sp)[0]
            XT = torch.empty_like(XT)
            ops.linear(hmid, pack([blk.mlp.fc2.weight], tag + ('fc2',)), C, prec, bias=blk.mlp.fc2.bias.detach()[None], out=XT, resid=XT2)
            if (i + 1) in self.select_list:
                taps.append(XT.view(B, N, C)[:, 1:].to(prec.adt).reshape(B * hw, C))
        xf, _, _ = ops.layernorm(XT, self.norm.weight.detach(), self.norm.bias.detach(), self.norm.eps, prec)
        taps.append(xf.view(B, N, C)[:, 1:].reshape(B * hw, C).contiguous())
        return taps


def _create_vision_transformer(variant, pretrained=False, default_cfg=None, **kwargs):
    kwargs.pop('representation_size', None)
    model = VisionTransformer(**kwargs)
    model.default_cfg = dict(default_cfg or {}, variant=variant)
    if pretrained:                       # vit.py:541: from the torch hub cache only (checkpoints.load_cached_pretrained), never the network
        from .checkpoints import load_cached_pretrained
        load_cached_pretrained(model, variant)
    return model


def vit_large_patch16_384(pretrained=False, **kwargs):
    """ViT-L/16 (vit.py:556-562)."""
    model_kwargs = dict(select_list=[6, 12, 18], patch_size=16, embed_dim=1024, depth=24, num_heads=16, **kwargs)
    return _create_vision_transformer('vit_large_patch16_384', pretrained=pretrained, **model_kwargs)


# ---------------------------------------------------------------------------------------------------------
# Decoder parameter holders (names / shapes of transformer_decoder.py and invpt.py)
# ---------------------------------------------------------------------------------------------------------
class UpEmbed(nn.Module):
    def __init__(self, in_chans, embed_dim):
        super().__init__()
        self.proj = nn.Sequential(nn.Upsample(scale_factor=2, mode='bilinear', align_corners=False),
                                  nn.Conv2d(in_chans, embed_dim, 3, padding=2, stride=1, bias=False, dilation=2), BATCHNORM(embed_dim),
                                  nn.ReLU(inplace=True),
                                  nn.Conv2d(embed_dim, embed_dim, 3, padding=2, stride=1, bias=False, dilation=2), BATCHNORM(embed_dim),
                                  nn.ReLU(inplace=True))


class InvPTMlp(nn.Module):
    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.fc2 = nn.Linear(hidden, dim)


class SelfAttention(nn.Module):
    def __init__(self, fea_no, dim, num_heads, qkv_bias=True):
        super().__init__()
        self.num_heads, self.dim, self.fea_no = num_heads, dim, fea_no
        self.conv_proj_q = nn.ModuleList([nn.Sequential(OrderedDict([
            ('conv', nn.Conv2d(dim, dim, 3, padding=1, stride=2, bias=False, groups=dim)), ('bn', BATCHNORM(dim))]))
            for _ in range(fea_no)])
        self.proj_q = nn.Linear(dim, dim, bias=qkv_bias)
        self.proj_k = nn.Linear(dim, dim, bias=qkv_bias)
        self.proj_v = nn.Linear(dim, dim, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim)
        self.fuse_attn = nn.Conv2d(num_heads * 2, num_heads, 1)


class InvPTBlock(nn.Module):
    def __init__(self, task_no, dim, num_heads, mlp_ratio=4.):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim)
        self.norm2 = nn.LayerNorm(dim)
        self.mlp = InvPTMlp(dim, int(dim * mlp_ratio))
        self.attn = SelfAttention(task_no, dim, num_heads)


def _init_trunc(m):
    if isinstance(m, nn.Linear):
        trunc_normal_(m.weight, std=0.02)
        if m.bias is not None:
            nn.init.constant_(m.bias, 0)
    elif isinstance(m, (nn.LayerNorm, nn.modules.batchnorm._BatchNorm)):
        nn.init.constant_(m.bias, 0)
        nn.init.constant_(m.weight, 1.0)


class InvPTStage(nn.Module):
    def __init__(self, task_no, stage_idx, in_chans, embed_dim, num_heads):
        super().__init__()
        self.stage_idx = stage_idx
        self.patch_embed = None if stage_idx == 0 else nn.ModuleList([UpEmbed(in_chans, embed_dim) for _ in range(task_no)])
        self.blocks = nn.ModuleList([InvPTBlock(task_no, embed_dim, num_heads)])
        self.apply(_init_trunc)


class InvPT(nn.Module):
    def __init__(self, p, in_chans, spec):
        super().__init__()
        self.p = p
        self.all_tasks = p.TASKS.NAMES
        T = len(self.all_tasks)
        self.norm_mts = nn.ModuleList()
        self.redu_chan = nn.ModuleList()
        self.invpt_stages = nn.ModuleList()
        self.mt_embed_dims = []
        cur = in_chans
        for i in range(spec['NUM_STAGES']):
            d = spec['DIM_EMBED'][i]
            self.invpt_stages.append(InvPTStage(T, i, cur, d, spec['NUM_HEADS'][i]))
            cur = d
            self.norm_mts.append(nn.LayerNorm(d * T))
            self.mt_embed_dims.append(d)
            self.redu_chan.append(nn.ModuleList([nn.Conv2d(d, in_chans, 1) for _ in range(T)]))
        self.norm_mt = nn.LayerNorm(T * cur)
        self.mt_proj = nn.ModuleDict()
        for task in self.all_tasks:
            self.mt_proj[task] = nn.Sequential(nn.Conv2d(in_chans, in_chans, 3, padding=1), BATCHNORM(in_chans), nn.ReLU(True))
            trunc_normal_(self.mt_proj[task][0].weight, std=0.02)
        self.mix_proj = nn.ModuleDict()
        for t in self.all_tasks:
            self.mix_proj[t] = nn.Sequential(nn.Conv2d(spec['ori_embed_dim'] + p.TASKS.NUM_OUTPUT[t], in_chans, 1))


class ConvBlock(nn.Module):
    def __init__(self, inplanes, planes):
        super().__init__()
        self.conv = nn.Conv2d(inplanes, planes, kernel_size=3, stride=1, padding=1, bias=False)
        self.bn1 = BATCHNORM(planes)
        self.relu = nn.ReLU(inplace=True)


class MLPHead(nn.Module):
    """transformer_decoder.py:124-131."""

    def __init__(self, in_channels, num_classes):
        super().__init__()
        self.linear_pred = nn.Conv2d(in_channels, num_classes, kernel_size=1)


def _fold(bns, conv_biases, tag):
    def build():
        with torch.no_grad():
            sc = torch.stack([bn.weight.detach() * torch.rsqrt(bn.running_var + bn.eps) for bn in bns])
            sh = torch.stack([bn.bias.detach() - bn.running_mean * s for bn, s in zip(bns, sc)])
            if conv_biases is not None:
                sh = sh + torch.stack([b.detach() for b in conv_biases]) * sc
            return sc.contiguous(), sh.contiguous()
    prm = [q for bn in bns for q in (bn.weight, bn.bias, bn.running_mean, bn.running_var)] + list(conv_biases or [])
    return ops._cached((tag, tuple(id(q) for q in prm)), prm, build)


def _bn_train(y, bns, C, act):
    """train-mode BatchNorm (+act) on [Z, rows, ld]; updates running statistics (momentum 0.1)."""
    from . import bn as bn_mod
    return bn_mod.train_forward(y, C, list(bns), act)[0]


class TransformerDecoder(nn.Module):
    """transformer_decoder.py:18-98 + InvPT.forward (invpt.py:502-544)."""

    def __init__(self, p):
        super().__init__()
        self.embed_dim = p.embed_dim
        E = self.embed_dim + p.PRED_OUT_NUM_CONSTANT
        p.mtt_resolution = [_ // p.mtt_resolution_downsample_rate for _ in p.spatial_dim[-1]]
        self.p = p
        spec = dict(ori_embed_dim=self.embed_dim, NUM_STAGES=3, DIM_EMBED=[E, E // 2, E // 4], NUM_HEADS=[2, 2, 2])
        input_channels = p.backbone_channels[-1]
        self.intermediate_head = nn.ModuleDict()
        self.invpt = InvPT(p, in_chans=E, spec=spec)
        self.preliminary_decoder = nn.ModuleDict()
        for t in p.TASKS.NAMES:
            self.intermediate_head[t] = nn.Conv2d(self.embed_dim, p.TASKS.NUM_OUTPUT[t], 1)
            self.preliminary_decoder[t] = nn.Sequential(ConvBlock(input_channels, input_channels), ConvBlock(input_channels, self.embed_dim))
        self.scale_embed = nn.ModuleList()
        self.scale_embed.append(nn.ConvTranspose2d(p.backbone_channels[0], E // 4, kernel_size=3, stride=2, padding=1, output_padding=1))
        self.scale_embed.append(nn.Conv2d(p.backbone_channels[1], E // 2, 3, padding=1))
        self.scale_embed.append(nn.Conv2d(p.backbone_channels[2], E, 3, padding=1))
        self.scale_embed.append(None)
        self.prec = _prec_of(p)

    # -------------------------------------------------------------------------------------------------
    def forward(self, x_list):
        """x_list: 4 token maps [B, hw, C] (reference API) -> ({task: [B, E, 8mh, 8mw]}, {task: [B, n, mh, mw]})."""
        B = x_list[0].shape[0]
        taps = [t.reshape(-1, t.shape[-1]).to(self.prec.adt).contiguous() for t in x_list]
        feats, inter = self.forward_nhwc(taps, B)
        names, p = self.p.TASKS.NAMES, self.p
        mh, mw = p.mtt_resolution
        E = self.embed_dim + p.PRED_OUT_NUM_CONSTANT
        out = {t: feats[i].view(B, 8 * mh, 8 * mw, -1)[..., :E].permute(0, 3, 1, 2).float() for i, t in enumerate(names)}
        ip = {t: inter[t].view(B, mh, mw, -1)[..., :p.TASKS.NUM_OUTPUT[t]].permute(0, 3, 1, 2).float() for t in names}
        return out, ip

    def forward_nhwc(self, taps, B):
        """taps: 4 contiguous [B*hw, C] maps.  -> (features [T, B*8mh*8mw, pitch(E)], {task: inter_pred [B*mh*mw, pitch(n)] fp32})."""
        if torch.is_grad_enabled() and (any(t.requires_grad for t in taps) or any(q.requires_grad for q in self.parameters())):
            from . import invpt_autograd
            f, inter = invpt_autograd.decoder_forward(self, taps, B)
            return f, {t: v[0] for t, v in inter.items()}
        p, prec = self.p, self.prec
        names = p.TASKS.NAMES
        T = len(names)
        h, w = p.spatial_dim[-1]
        mh, mw = p.mtt_resolution
        C = p.backbone_channels[-1]
        Ed = self.embed_dim
        E = Ed + p.PRED_OUT_NUM_CONSTANT
        dims = [E, E // 2, E // 4]
        training = self.training
        dev = taps[0].device

        # ---- multi-scale skip features (scale_embed[2] is dead in the reference: skipped) -------------------
        se0, se1 = self.scale_embed[0], self.scale_embed[1]
        Co0, Co0p = dims[2], pitch(dims[2])

        def build_wall():
            with torch.no_grad():
                buf = torch.zeros(9, Co0p, C, dtype=torch.float32, device=dev)
                buf[:, :Co0] = se0.weight.detach().permute(2, 3, 1, 0).reshape(9, Co0, C)      # [ci, co, ky, kx] -> [tap, co, ci]
                return ops.pack_matrix(buf.reshape(9 * Co0p, C), prec)[None]
        wall = ops._cached(('se0', prec.name, id(se0.weight)), [se0.weight], build_wall)
        yall = ops.linear(taps[0], wall, 9 * Co0p, prec, ldd=9 * Co0p)[0]
        bias0 = ops._cached(('se0b', id(se0.bias)), [se0.bias],
                            lambda: torch.cat([se0.bias.detach(), se0.bias.new_zeros(Co0p - Co0)]).contiguous())
        back0 = torch.empty(B * 4 * h * w, Co0p, dtype=prec.adt, device=dev)
        ops.call("convt3x3s2_gather", yall=yall, out=back0, bias=bias0, B=B, H=h, W=w, Cop=Co0p, dtype=dtype_code(yall),
                 out_dtype=dtype_code(back0))
        back1 = ops.conv3x3(taps[1][None], ops.pack_conv3([se1.weight], prec, 'se1'), dims[1], C, B, h, w, prec,
                            bias=se1.bias.detach()[None].contiguous())[0]

        # ---- preliminary decoder + intermediate heads (transformer_decoder.py:85-95) --------------------------
        x = ops.bilinear(taps[3][None], B, C, h, w, mh, mw, prec.adt)                       # [1, B*mh*mw, C]
        rows0 = B * mh * mw
        pd = [self.preliminary_decoder[t] for t in names]
        W0 = ops.pack_conv3([m[0].conv.weight for m in pd], prec, 'pd0')
        W1 = ops.pack_conv3([m[1].conv.weight for m in pd], prec, 'pd1')
        xin = x.expand(T, rows0, x.shape[-1])
        if training:
            y = _bn_train(ops.conv3x3(xin, W0, C, C, B, mh, mw, prec), [m[0].bn1 for m in pd], C, ACT_RELU)
            y = _bn_train(ops.conv3x3(y, W1, Ed, C, B, mh, mw, prec), [m[1].bn1 for m in pd], Ed, ACT_RELU)
        else:
            sc, sh = _fold([m[0].bn1 for m in pd], None, 'pd0bn')
            y = ops.conv3x3(xin, W0, C, C, B, mh, mw, prec, bias=sh, colscale=sc, act=ACT_RELU)
            sc, sh = _fold([m[1].bn1 for m in pd], None, 'pd1bn')
            y = ops.conv3x3(y, W1, Ed, C, B, mh, mw, prec, bias=sh, colscale=sc, act=ACT_RELU)
        Edp = pitch(Ed)
        inter, xs = {}, []
        for i, t in enumerate(names):
            n_out = p.TASKS.NUM_OUTPUT[t]
            ih = self.intermediate_head[t]
            inter[t] = ops.linear(y[i], ops.pack_linear([ih.weight], prec, ('ih', t)), n_out, prec, bias=ih.bias.detach()[None],
                                  out_dtype=torch.float32)[0]
            # mix_proj on cat([feature, inter_pred]) (invpt.py:509-513): two GEMMs, the second accumulates
            mp = self.invpt.mix_proj[t][0]
            wa = ops._cached(('mixa', t, prec.name, id(mp.weight)), [mp.weight],
                             lambda mp=mp: ops.pack_matrix(mp.weight.detach().reshape(E, -1)[:, :Ed].contiguous(), prec)[None])
            wb = ops._cached(('mixb', t, prec.name, id(mp.weight)), [mp.weight],
                             lambda mp=mp: ops.pack_matrix(mp.weight.detach().reshape(E, -1)[:, Ed:].contiguous(), prec)[None])
            part = ops.linear(y[i], wa, E, prec, bias=mp.bias.detach()[None], out_dtype=torch.float32)[0]
            xs.append(ops.linear(inter[t], wb, E, prec, resid=part)[0])      # f32 A operand is converted while staging
        X = torch.stack(xs, 0)                                                              # [T, rows0, pitch(E)]

        # ---- InvPT stages --------------------------------------------------------------------------------------
        th, tw = mh * 8, mw * 8
        Ep = pitch(E)
        acc = torch.zeros(T, B * th * tw, Ep, dtype=torch.float32, device=dev)
        prev_score = None
        gh, gw = mh, mw
        for i in range(3):
            D, Dp = dims[i], pitch(dims[i])
            stage = self.invpt.invpt_stages[i]
            blk = stage.blocks[0]
            if i > 0:
                ue = [m.proj for m in stage.patch_embed]
                Din = dims[i - 1]
                up = ops.bilinear(X, B, X.shape[-1], gh, gw, 2 * gh, 2 * gw, prec.adt)
                gh, gw = 2 * gh, 2 * gw
                Wc1 = ops.pack_conv3([m[1].weight for m in ue], prec, ('ue1', i))
                Wc2 = ops.pack_conv3([m[4].weight for m in ue], prec, ('ue2', i))
                if training:
                    yy = _bn_train(ops.conv3x3(up, Wc1, D, Din, B, gh, gw, prec, dil=2), [m[2] for m in ue], D, ACT_RELU)
                    yy = _bn_train(ops.conv3x3(yy, Wc2, D, D, B, gh, gw, prec, dil=2), [m[5] for m in ue], D, ACT_RELU)
                else:
                    sc, sh = _fold([m[2] for m in ue], None, ('ue1bn', i))
                    yy = ops.conv3x3(up, Wc1, D, Din, B, gh, gw, prec, dil=2, bias=sh, colscale=sc, act=ACT_RELU)
                    sc, sh = _fold([m[5] for m in ue], None, ('ue2bn', i))
                    yy = ops.conv3x3(yy, Wc2, D, D, B, gh, gw, prec, dil=2, bias=sh, colscale=sc, act=ACT_RELU)
                skip = back1 if i == 1 else back0                                           # invpt.py:406-411
                X = yy
                Xf = torch.empty(T, B * gh * gw, Dp, dtype=torch.float32, device=dev)
                for t in range(T):
                    ops.call("cast2d", args=[X[t], Xf[t], B * gh * gw, Dp, Dp, Dp, dtype_code(X), F32, 0])
                    ops.call("add_rows", args=[skip, Xf[t], B * gh * gw, Dp, Dp, Dp, dtype_code(skip), 1.0])
            else:
                Xf = X.float() if X.dtype != torch.float32 else X
            rows = B * gh * gw
            Xf, prev_score = self._block(blk, i, Xf, B, T, D, gh, gw, prev_score)
            # LayerNorm over all tasks' channels, redu_chan (i > 0), resize to the target grid and accumulate
            yn = torch.empty(T, rows, Dp, dtype=prec.adt, device=dev)
            nm = self.invpt.norm_mts[i]
            ops.call("layernorm_mt", x=Xf, y=yn, gamma=nm.weight.detach(), beta=nm.bias.detach(), rows=rows, T=T, D=D, ldx=Dp, ldy=Dp,
                     y_dtype=dtype_code(yn), eps=nm.eps)
            if i > 0:
                rc = self.invpt.redu_chan[i]
                yn = ops.linear(yn, ops.pack_linear([m.weight for m in rc], prec, ('rc', i)), E, prec,
                                bias=ops.stack_vec([m.bias for m in rc], ('rcb', i)))
            ops.call("bilinear_fwd", **{"in": yn}, out=acc, B=T * B, C=Ep, Hin=gh, Win=gw, Hout=th, Wout=tw, ld_in=Ep, ld_out=Ep,
                     in_dtype=dtype_code(yn), out_dtype=F32, out_nchw=0, accumulate=1)
            X = Xf
        # ---- mt_proj: 3x3 conv + BN + ReLU at the target resolution -----------------------------------------------
        mps = [self.invpt.mt_proj[t] for t in names]
        accq = acc if prec.adt == torch.float32 else ops.cast2d(acc.view(-1, Ep), acc.shape[0] * acc.shape[1], Ep, Ep, prec.adt, ldd=Ep).view(acc.shape)
        Wm = ops.pack_conv3([m[0].weight for m in mps], prec, 'mtp')
        if training:
            f = ops.conv3x3(accq, Wm, E, E, B, th, tw, prec, bias=ops.stack_vec([m[0].bias for m in mps], 'mtpb'))
            f = _bn_train(f, [m[1] for m in mps], E, ACT_RELU)
        else:
            sc, sh = _fold([m[1] for m in mps], [m[0].bias for m in mps], 'mtpbn')
            f = ops.conv3x3(accq, Wm, E, E, B, th, tw, prec, bias=sh, colscale=sc, act=ACT_RELU)
        return f, inter

    # -------------------------------------------------------------------------------------------------
    def _block(self, blk, si, Xf, B, T, D, gh, gw, prev_score):
        """InvPTBlock (invpt.py:290-312) on task-major fp32 tokens Xf [T, B*g*g, Dp].  Returns (new Xf, attention scores)."""
        prec = self.prec
        at = blk.attn
        heads = at.num_heads
        Dp = Xf.shape[-1]
        rows = B * gh * gw
        dev = Xf.device
        hd = D // heads
        hdp = pitch(hd)
        Dh = heads * hdp
        qh, qw = (gh - 1) // 2 + 1, (gw - 1) // 2 + 1
        kk = 2 ** (si + 1)
        kh_, kw_ = -(-gh // kk), -(-gw // kk)
        nq, nk = qh * qw, kh_ * kw_
        Q, K = T * nq, T * nk
        Kp = pitch(K)
        tag = ('ipb', si)
        xn = self._ln_padded(Xf, blk.norm1, D)
        # queries: depthwise 3x3 stride-2 conv + BN per task; keys / values: ceil-mode average pooling
        wq = ops._cached(('dwq', si, tuple(id(m.conv.weight) for m in at.conv_proj_q)), [m.conv.weight for m in at.conv_proj_q],
                         lambda: self._pack_dw([m.conv.weight for m in at.conv_proj_q], Dp))
        qmap = torch.empty(T, B * nq, Dp, dtype=prec.adt, device=dev)
        bns = [m.bn for m in at.conv_proj_q]
        if self.training:
            ops.call("dwconv3x3s2", x=xn, w=wq, y=qmap, scale=None, shift=None, Z=T, B=B, H=gh, W=gw, ld=Dp, dtype=dtype_code(xn))
            qmap = _bn_train(qmap, bns, D, ACT_NONE)
        else:
            sc, sh = _fold(bns, None, ('dwqbn', si))
            scp, shp = self._pad_cols(sc, Dp), self._pad_cols(sh, Dp)
            ops.call("dwconv3x3s2", x=xn, w=wq, y=qmap, scale=scp, shift=shp, Z=T, B=B, H=gh, W=gw, ld=Dp, dtype=dtype_code(xn))
        kvmap = torch.empty(T, B * nk, Dp, dtype=prec.adt, device=dev)
        ops.call("avgpool_ceil", x=xn, y=kvmap, B=T * B, H=gh, W=gw, k=kk, ld=Dp, dtype=dtype_code(xn))

        # shared-weight projections; outputs batch-major [B, T*n, heads*hdp] through the row-group output mapping
        def proj(src, lin, n, name):
            wpk, bpk = self._pack_heads(lin, D, heads, hd, hdp, tag + (name,), prec)
            out = torch.empty(B, T * n, Dh, dtype=prec.adt, device=dev)
            kw = dict(A=src, B=wpk, D=out, M=B * n, N=Dh, K=wpk.shape[-1], a_op=OP_K, b_op=OP_K, a_dtype=dtype_code(src),
                      b_dtype=dtype_code(wpk), d_dtype=dtype_code(out), prec=prec.code, lda=Dp, ldb=wpk.shape[-1], ldd=Dh,
                      d_mb=n, d_bs=T * n * Dh, batch=T, batch_inner=1, a_zo=B * n * Dp, b_zo=0, d_zo=n * Dh, alpha=1.0,
                      colshift=bpk, n_store=Dh)
            ops.call("gemm", **kw)
            return out
        q = proj(qmap, at.proj_q, nq, 'q')
        k = proj(kvmap, at.proj_k, nk, 'k')
        v = proj(kvmap, at.proj_v, nk, 'v')
        # scores S[b, head] = scale * q k^T  (scale = D^-0.5: the FULL dim, invpt.py:92)
        Z = B * heads
        S = torch.empty(B, heads, Q, Kp, dtype=torch.float32, device=dev)
        ops.call("gemm", A=q, B=k, D=S, M=Q, N=K, K=hdp, a_op=OP_K, b_op=OP_K, a_dtype=dtype_code(q), b_dtype=dtype_code(k), d_dtype=F32,
                 prec=prec.code, lda=Dh, ldb=Dh, ldd=Kp, batch=Z, batch_inner=heads, a_zo=Q * Dh, a_zi=hdp, b_zo=K * Dh, b_zi=hdp,
                 d_zo=heads * Q * Kp, d_zi=Q * Kp, alpha=float(D) ** -0.5, n_store=Kp)
        if prev_score is not None:
            fa = at.fuse_attn
            S2 = torch.empty_like(S)
            ops.call("attn_msg", cur=S, prev=prev_score, out=S2, w=fa.weight.detach().reshape(heads, 2 * heads).contiguous(),
                     bias=fa.bias.detach(), B=B, heads=heads, T=T, qh=qh, qw=qw, K=K, ldk=Kp, ldkp=prev_score.shape[-1])
            S = S2
        P = torch.empty(B, heads, Q, Kp, dtype=prec.adt, device=dev)
        ops.call("softmax_fwd", S=S, P=P, rows=Z * Q, cols=K, ld=Kp, s_dtype=F32, p_dtype=dtype_code(P), scale=1.0)
        o = torch.empty(B, Q, Dh, dtype=prec.adt, device=dev)
        ops.call("gemm", A=P, B=v, D=o, M=Q, N=hdp, K=K, a_op=OP_K, b_op=OP_R, a_dtype=dtype_code(P), b_dtype=dtype_code(v),
                 d_dtype=dtype_code(o), prec=prec.code, lda=Kp, ldb=Dh, ldd=Dh, batch=Z, batch_inner=heads, a_zo=heads * Q * Kp, a_zi=Q * Kp,
                 b_zo=K * Dh, b_zi=hdp, d_zo=Q * Dh, d_zi=hdp, alpha=1.0, n_store=hdp)
        # output projection back to task-major maps at the query resolution
        wpo = ops._cached(tag + ('po', prec.name, id(at.proj.weight)), [at.proj.weight],
                          lambda: self._pack_head_cols(at.proj.weight, D, heads, hd, hdp, prec))
        om = torch.empty(T, B * nq, Dp, dtype=prec.adt, device=dev)
        ops.call("gemm", A=o, B=wpo, D=om, M=B * nq, N=D, K=Dh, a_op=OP_K, b_op=OP_K, a_dtype=dtype_code(o), b_dtype=dtype_code(wpo),
                 d_dtype=dtype_code(om), prec=prec.code, lda=Dh, ldb=wpo.shape[-1], ldd=Dp, a_mb=nq, a_bs=T * nq * Dh, batch=T, batch_inner=1,
                 a_zo=nq * Dh, b_zo=0, d_zo=B * nq * Dp, alpha=1.0, colshift=at.proj.bias.detach(), n_store=Dp)
        # bilinear upsample of the attention output to the stage grid + residual (invpt.py:300-307)
        X2 = Xf.clone()
        ops.call("bilinear_fwd", **{"in": om}, out=X2, B=T * B, C=Dp, Hin=qh, Win=qw, Hout=gh, Wout=gw, ld_in=Dp, ld_out=Dp,
                 in_dtype=dtype_code(om), out_dtype=F32, out_nchw=0, accumulate=1)
        # MLP (shared weights) with residual
        xn2 = self._ln_padded(X2, blk.norm2, D)
        Hd = blk.mlp.fc1.weight.shape[0]
        hmid = ops.linear(xn2.view(T * rows, Dp), ops.pack_linear([blk.mlp.fc1.weight], prec, tag + ('fc1',)), Hd, prec,
                          bias=blk.mlp.fc1.bias.detach()[None], act=ACT_GELU)[0]
        X3 = torch.empty(T * rows, Dp, dtype=torch.float32, device=dev)
        ops.linear(hmid, ops.pack_linear([blk.mlp.fc2.weight], prec, tag + ('fc2',)), D, prec, bias=blk.mlp.fc2.bias.detach()[None],
                   out=X3, resid=X2.view(T * rows, Dp), n_store=Dp)
        return X3.view(T, rows, Dp), S

    def _ln_padded(self, Xf, norm, D):
        """LayerNorm over the D valid channels of fp32 [T, rows, Dp] -> activation dtype [T*rows, Dp] with zero padding."""
        T, rows, Dp = Xf.shape
        prec = self.prec
        y = torch.zeros(T * rows, Dp, dtype=prec.adt, device=Xf.device) if Dp != D else torch.empty(T * rows, Dp, dtype=prec.adt, device=Xf.device)
        if D % 4 == 0:
            ops.call("layernorm_fwd", x=Xf, y=y, gamma=norm.weight.detach(), beta=norm.bias.detach(), mean=None, rstd=None,
                     rows=T * rows, C=D, ldx=Dp, ldy=Dp, y_dtype=dtype_code(y), eps=norm.eps)
        else:   # odd channel counts (miniature configs): the multi-task LN kernel with T = 1 handles any D
            ops.call("layernorm_mt", x=Xf, y=y, gamma=norm.weight.detach(), beta=norm.bias.detach(), rows=T * rows, T=1, D=D, ldx=Dp, ldy=Dp,
                     y_dtype=dtype_code(y), eps=norm.eps)
        return y.view(T, rows, Dp)

    @staticmethod
    def _pad_cols(t2d, Dp):
        if t2d.shape[-1] == Dp:
            return t2d
        out = torch.zeros(t2d.shape[0], Dp, dtype=t2d.dtype, device=t2d.device)
        out[:, :t2d.shape[-1]] = t2d
        return out

    @staticmethod
    def _pack_dw(ws, Dp):
        with torch.no_grad():
            out = torch.zeros(len(ws), 9, Dp, dtype=torch.float32, device=ws[0].device)
            for z, wt in enumerate(ws):
                out[z, :, :wt.shape[0]] = wt.detach().reshape(wt.shape[0], 9).t()
            return out

    @staticmethod
    def _pack_heads(lin, D, heads, hd, hdp, tag, prec):
        """Linear [D, D] -> rows re-laid as heads padded to hdp: [heads*hdp, pitch(D)] (+ matching bias)."""
        def build():
            with torch.no_grad():
                Wt = torch.zeros(heads * hdp, D, dtype=torch.float32, device=lin.weight.device)
                bt = torch.zeros(heads * hdp, dtype=torch.float32, device=lin.weight.device)
                for hh in range(heads):
                    Wt[hh * hdp:hh * hdp + hd] = lin.weight.detach()[hh * hd:(hh + 1) * hd]
                    if lin.bias is not None:
                        bt[hh * hdp:hh * hdp + hd] = lin.bias.detach()[hh * hd:(hh + 1) * hd]
                return ops.pack_matrix(Wt, prec), bt
        return ops._cached(tag + (prec.name, id(lin.weight)), [lin.weight] + ([lin.bias] if lin.bias is not None else []), build)

    @staticmethod
    def _pack_head_cols(weight, D, heads, hd, hdp, prec):
        with torch.no_grad():
            Wt = torch.zeros(D, heads * hdp, dtype=torch.float32, device=weight.device)
            for hh in range(heads):
                Wt[:, hh * hdp:hh * hdp + hd] = weight.detach()[:, hh * hd:(hh + 1) * hd]
            return ops.pack_matrix(Wt, prec)


class TransformerNet(nn.Module):
    """transformer_net.py:13-38."""

    def __init__(self, p, backbone, backbone_channels, heads):
        super().__init__()
        self.tasks = p.TASKS.NAMES
        self.backbone = backbone
        self.multi_task_decoder = TransformerDecoder(p)
        self.heads = heads
        self.p = p

    def forward(self, x):
        if torch.is_grad_enabled() and any(q.requires_grad for q in self.parameters()):
            from . import invpt_autograd
            return invpt_autograd.net_forward(self, x)
        img_size = tuple(x.shape[-2:])
        B = x.shape[0]
        dec = self.multi_task_decoder
        prec = dec.prec
        taps = self.backbone.forward_taps(x)
        feats, inter = dec.forward_nhwc(taps, B)
        mh, mw = self.p.mtt_resolution
        th, tw = 8 * mh, 8 * mw
        out = {}
        for i, t in enumerate(self.tasks):
            hd = self.heads[t]
            n_out = hd.linear_pred.weight.shape[0]
            pred = ops.linear(feats[i], ops.pack_linear([hd.linear_pred.weight], prec, ('iph', t)), n_out, prec,
                              bias=hd.linear_pred.bias.detach()[None], out_dtype=torch.float32)
            out[t] = ops.bilinear(pred, B, n_out, th, tw, img_size[0], img_size[1], torch.float32, nchw=True)
        out['inter_preds'] = {t: ops.bilinear(inter[t][None], B, self.p.TASKS.NUM_OUTPUT[t], mh, mw, img_size[0], img_size[1],
                                              torch.float32, nchw=True) for t in self.tasks}
        return out
