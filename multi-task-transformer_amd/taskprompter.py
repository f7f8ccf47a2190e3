"""MI355X-native TaskPrompter: same nn.Module API, constructor arguments and state_dict layout as
the reference (TaskPrompter/models/transformers/taskprompter.py, TaskPrompter/models/taskprompter_wrapper.py)
so it drops into the reference's get_model / main.py / inference.py, but executed as a different, fused
schedule on the libmtt_hip.so kernels:

  * one token-major fp32 residual buffer XT [B, T+hw, C] holds prompts (first, as taskprompter.py:199)
    and patches; norm1/norm2/MLP run once over all rows (the reference applies the same weights to x and
    to the prompts separately, :272-277)
  * LayerNorm -> qkv GEMM -> flash attention that also emits the T prompt-row logits (the only rows
    cal_task_feature reads) -> proj GEMM with the residual add in its epilogue
  * channel attention = token_trans GEMM + a coalesced logit kernel; the dead softmax / attn@v of
    taskprompter.py:241-245 are skipped (parameters keep their reference names)
  * cal_task_feature = one modulation kernel + task-batched GEMMs / implicit-GEMM 3x3 convs with bias,
    eval-BatchNorm and GELU folded into epilogues, cross-task mixing fused with the 4-tap accumulation
  * heads = x4 bilinear (NHWC) -> task-batched 3x3 conv + BN + GELU -> 1x1 -> bilinear to NCHW fp32

The nn.Linear / nn.Conv2d / nn.BatchNorm2d / nn.LayerNorm children are parameter holders only (so
`SyncBatchNorm.convert_sync_batchnorm`, `load_state_dict(strict=True)` and optimizers see the reference's
parameters); their forward is never called.
"""
import math
from functools import partial

import torch
import torch.nn as nn

from . import ops
from ._lib import ACT_GELU, ACT_NONE

BatchNorm2d = nn.BatchNorm2d
INTERPOLATE_MODE = 'bilinear'


def trunc_normal_(tensor, mean=0., std=1., a=-2., b=2.):
    """timm 0.5.4 semantics (absolute cut-offs), used for init parity of freshly constructed models."""
    with torch.no_grad():
        lo = (1. + math.erf((a - mean) / std / math.sqrt(2.))) / 2.
        hi = (1. + math.erf((b - mean) / std / math.sqrt(2.))) / 2.
        tensor.uniform_(2 * lo - 1, 2 * hi - 1).erfinv_().mul_(std * math.sqrt(2.)).add_(mean).clamp_(min=a, max=b)
    return tensor


def _prec_of(p):
    return ops.Prec(p.get('mtt_prec', ops.DEFAULT_PREC) if hasattr(p, 'get') else getattr(p, 'mtt_prec', ops.DEFAULT_PREC))


PREC_GROUPS = ('enc', 'attn', 'side', 'fuse', 'heads')


def _group_precs(p, base):
    """Per-group OPERAND precision for the error-attribution runs (tools/prec_attribution.py): `p.mtt_prec_groups` maps a group
    of PREC_GROUPS to 'bf16' | 'x3'.  Storage stays fp32 (requires mtt_prec == 'x3'); a 'bf16' group rounds its GEMM / attention
    operands to bf16 exactly where the bf16 mode does (weights packed bf16, activations converted while staged), so the table
    isolates which part of the path the bf16 mode's error comes from.  Inference path only."""
    g = p.get('mtt_prec_groups', None) if hasattr(p, 'get') else getattr(p, 'mtt_prec_groups', None)
    if not g:
        return None
    assert base.name == 'x3', "mtt_prec_groups needs fp32 storage (mtt_prec = 'x3')"
    assert set(g) <= set(PREC_GROUPS), g
    return {k: ops.Prec(g.get(k, 'x3')) for k in PREC_GROUPS}


class PatchEmbed(nn.Module):
    """Parameter holder with timm's PatchEmbed attribute surface (proj, grid_size, num_patches)."""

    def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768):
        super().__init__()
        self.img_size = tuple(img_size)
        self.patch_size = (patch_size, patch_size)
        self.grid_size = (img_size[0] // patch_size, img_size[1] // patch_size)
        self.num_patches = self.grid_size[0] * self.grid_size[1]
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size)


class Mlp(nn.Module):
    def __init__(self, in_features, hidden_features):
        super().__init__()
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.fc2 = nn.Linear(hidden_features, in_features)


class Attention(nn.Module):
    """Parameters of taskprompter.py:168-193 (qkv, proj, token_trans [hw<-C], token_trans1 [C<-hw])."""

    def __init__(self, chan_nheads, resolution, dim, num_heads=8, qkv_bias=False):
        super().__init__()
        self.num_heads = num_heads
        self.dim = dim
        self.resolution = resolution
        self.pixel_no = int(resolution[0] * resolution[1])
        self.chan_nheads = chan_nheads
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim)
        self.token_trans = nn.Linear(dim, self.pixel_no)
        self.token_trans1 = nn.Linear(self.pixel_no, dim)


class Block(nn.Module):
    def __init__(self, chan_nheads, resolution, dim, num_heads, mlp_ratio=4., qkv_bias=False, drop_path=0.,
                 norm_layer=nn.LayerNorm):
        super().__init__()
        self.norm1 = norm_layer(dim)
        self.attn = Attention(chan_nheads, resolution, dim, num_heads=num_heads, qkv_bias=qkv_bias)
        self.drop_path_rate = float(drop_path)
        self.norm2 = norm_layer(dim)
        self.mlp = Mlp(dim, int(dim * mlp_ratio))


def _init_vit_weights(m):
    if isinstance(m, nn.Linear):
        trunc_normal_(m.weight, std=.02)
        if m.bias is not None:
            nn.init.zeros_(m.bias)
    elif isinstance(m, (nn.LayerNorm, nn.BatchNorm2d)):
        nn.init.zeros_(m.bias)
        nn.init.ones_(m.weight)


class TaskPrompter(nn.Module):
    """TaskPrompter built upon ViT (taskprompter.py:281-487)."""

    def __init__(self, p, select_list, img_size=224, patch_size=16, in_chans=3, embed_dim=768, depth=12,
                 num_heads=12, chan_nheads=1, mlp_ratio=4., qkv_bias=True, drop_rate=0., attn_drop_rate=0.,
                 drop_path_rate=0., embed_layer=None, norm_layer=None, act_layer=None, weight_init=''):
        super().__init__()
        assert patch_size == 16 and in_chans == 3 and embed_dim % 64 == 0 and embed_dim // num_heads == 64, \
            "HIP path: patch 16, head_dim 64"
        assert drop_rate == 0. and attn_drop_rate == 0., "reference never enables these (taskprompter.py:318)"
        self.num_features = self.embed_dim = embed_dim
        norm_layer = norm_layer or partial(nn.LayerNorm, eps=1e-6)
        if isinstance(img_size, int):
            img_size = (img_size, img_size)
        self.patch_embed = PatchEmbed(img_size=img_size, patch_size=patch_size, in_chans=in_chans, embed_dim=embed_dim)
        num_patches = self.patch_embed.num_patches
        self.pos_embed = nn.Parameter(torch.zeros(1, num_patches + 1, embed_dim))
        dpr = [x.item() for x in torch.linspace(0, drop_path_rate, depth)]
        self.resolution = [int(img_size[0] / patch_size), int(img_size[1] / patch_size)]
        self.blocks = nn.Sequential(*[
            Block(chan_nheads, self.resolution, dim=embed_dim, num_heads=num_heads, mlp_ratio=mlp_ratio,
                  qkv_bias=qkv_bias, drop_path=dpr[i], norm_layer=norm_layer) for i in range(depth)])
        self.norm = norm_layer(embed_dim)
        self.select_list = list(select_list)
        self.num_layers = 4
        assert len(self.select_list) == self.num_layers - 1
        self.num_heads = num_heads
        self.chan_nheads = chan_nheads
        task_no = len(p.TASKS.NAMES)
        self.pixel_no = int(self.resolution[0] * self.resolution[1])
        self.p = p
        self.prompt_len = p.prompt_len
        assert self.prompt_len == 1, "HIP path implements prompt_len == 1 (every reference config)"
        self.prompts_len = task_no * p.prompt_len
        self.task_prompts = nn.Parameter(torch.ones(self.prompts_len, embed_dim))
        trunc_normal_(self.task_prompts, mean=1., std=1.)

        self.fea_fuse = nn.ModuleList()
        if p.use_ctr:
            self.ctr_attn_conv = nn.ModuleList()
        self.fea_decode_spa = nn.ModuleList()
        self.fea_decode_chan = nn.ModuleList()
        prompt_dim = num_heads * p.prompt_len
        tar_dim = p.embed_dim
        F = p.final_embed_dim
        for _ in range(self.num_layers):
            self.fea_fuse.append(nn.ModuleDict())
            if p.use_ctr:
                self.ctr_attn_conv.append(nn.ModuleDict())
            self.fea_decode_spa.append(nn.ModuleDict())
            self.fea_decode_chan.append(nn.ModuleDict())
            for task in p.TASKS.NAMES:
                self.fea_fuse[-1][task] = nn.Sequential(nn.Conv2d(tar_dim * 2, F, kernel_size=1),
                                                        nn.Conv2d(F, F, kernel_size=3, padding=1), BatchNorm2d(F),
                                                        nn.GELU(), nn.Conv2d(F, F, kernel_size=1))
                if p.use_ctr:
                    self.ctr_attn_conv[-1][task] = nn.Sequential(nn.Conv2d(prompt_dim, prompt_dim, kernel_size=1),
                                                                 nn.GELU(), nn.Conv2d(prompt_dim, 1, kernel_size=1))
                self.fea_decode_spa[-1][task] = nn.Sequential(nn.Conv2d(embed_dim, tar_dim, kernel_size=1))
                self.fea_decode_chan[-1][task] = nn.Sequential(nn.Conv2d(embed_dim, tar_dim, kernel_size=1))
        trunc_normal_(self.pos_embed, std=.02)
        self.apply(_init_vit_weights)
        self.prec = _prec_of(p)
        self.gprec = _group_precs(p, self.prec)

    def _gp(self, group):
        """compute precision of a group (== the model's, unless an attribution run overrides it)"""
        return self.gprec[group] if self.gprec else self.prec

    @torch.jit.ignore
    def no_weight_decay(self):
        return {'pos_embed', 'cls_token', 'dist_token'}

    # ------------------------------------------------------------------------------------------
    def _tap_index(self, idx):
        return self.select_list.index(idx + 1)

    def _side_channels_used(self, idx):
        """are block idx's logit side channels (rawlog / rawchan) read by cal_task_feature?  Only the tap blocks' and the last block's are
        (taskprompter.py:405-417); the other blocks still compute the channel attention's prompt update, but its raw logits — one more
        pass over the block's tokens — would be dead compute."""
        return (idx + 1) in self.select_list or idx == len(self.blocks) - 1

    def load_pretrained(self, checkpoint_path, prefix=''):
        """taskprompter.py:385-386: import a Google/Flax ViT .npz into the encoder (prompts / decoders keep their initialisation)."""
        from .checkpoints import load_flax_vit_npz
        return load_flax_vit_npz(self, checkpoint_path, prefix)

    def forward(self, x):
        """-> ({task: [B, F, 4h, 4w]} (channels-last views of the NHWC buffers), info) as taskprompter.py:392-422."""
        fea = self.forward_nhwc(x)
        B = x.shape[0]
        F = self.p.final_embed_dim
        h4, w4 = self.resolution[0] * 4, self.resolution[1] * 4
        out = {t: fea[i].view(B, h4, w4, -1)[..., :F].permute(0, 3, 1, 2) for i, t in enumerate(self.p.TASKS.NAMES)}
        return out, {}

    def forward_nhwc(self, img, upsample=True):
        """-> [T, B*4h*4w, pitch(F)] activation-dtype task features (x4-upsampled sum over the 4 taps; taskprompter.py:420).
        upsample=False stops before the resize: fp32 [T, B*h*w, pitch(F)], for heads that fuse it into their first conv."""
        if torch.is_grad_enabled() and any(q.requires_grad for q in self.parameters()):
            from . import autograd_path
            return autograd_path.backbone_forward(self, img, upsample)
        return self._forward_nograd(img, upsample)

    def upsample4(self, acc, B):
        """the x4 bilinear resize of forward_nhwc(upsample=False)'s result (differentiable when that is)."""
        h, w = self.resolution
        if torch.is_grad_enabled() and acc.requires_grad:
            from . import autograd_path
            return autograd_path.upsample4(acc, B, h, w, self.prec)
        return ops.bilinear(acc, B, acc.shape[-1], h, w, 4 * h, 4 * w, self.prec.adt)

    def _forward_nograd(self, img, upsample=True):
        p, prec = self.p, self.prec
        B = img.shape[0]
        H, W = img.shape[-2:]
        assert (H, W) == tuple(self.patch_embed.img_size), "input size must equal img_size (timm PatchEmbed assert)"
        h, w = self.resolution
        hw, T, C, nH = h * w, self.prompts_len, self.embed_dim, self.num_heads
        N = T + hw
        dev = img.device
        nwin = int(math.isqrt(self.chan_nheads))
        assert h % nwin == 0 and w % nwin == 0

        # ---- patch embed + pos embed straight into the token buffer; prompts first ----------------
        XT = torch.empty(B * N, C, dtype=torch.float32, device=dev)
        XT.view(B, N, C)[:, :T] = self.task_prompts.detach()
        cols = ops.patchify(img.float(), prec)
        pe = self._gp('enc')
        wpe = ops.pack_linear([self.patch_embed.proj.weight], pe, 'pe')
        ops.linear(cols, wpe, C, pe, bias=self.patch_embed.proj.bias.detach()[None], out=XT.view(B, N, C)[:, T:],
                   d_rows=(hw, N * C, C), resid=self.pos_embed.detach()[0, 1:], r_rows=(hw, 0, C), M=B * hw)

        acc = None
        rawlog = rawchan = None
        for i, blk in enumerate(self.blocks):
            XT, rawlog, rawchan = self._block(blk, i, XT, B, N, T, (h, w), nwin)
            if (i + 1) in self.select_list:
                acc = self._task_features(XT, XT.view(B, N, C)[:, T:], rawlog, rawchan, self._tap_index(i), B, acc)
        xf, _, _ = ops.layernorm(XT, self.norm.weight.detach(), self.norm.bias.detach(), self.norm.eps, prec,
                                 out_dtype=torch.float32)
        acc = self._task_features(xf, xf.view(B, N, C)[:, T:], rawlog, rawchan, 3, B, acc)
        return ops.bilinear(acc, B, acc.shape[-1], h, w, 4 * h, 4 * w, prec.adt) if upsample else acc

    def _block(self, blk, i, XT, B, N, T, grid, nwin):
        prec, C, nH = self.prec, self.embed_dim, self.num_heads
        pe, pa, ps = self._gp('enc'), self._gp('attn'), self._gp('side')
        adt = prec.adt                                              # storage dtype (a group override only changes operand rounding)
        hw = grid[0] * grid[1]
        a = blk.attn
        tag = ('blk', i)
        if prec.split and self.gprec is None:
            return self._block_split(blk, tag, XT, B, N, T, grid, nwin, self._side_channels_used(i))
        xn, _, _ = ops.layernorm(XT, blk.norm1.weight.detach(), blk.norm1.bias.detach(), blk.norm1.eps, prec)
        qkv = ops.linear(xn, ops.pack_linear([a.qkv.weight], pe, tag + ('qkv',)), 3 * C, pe,
                         bias=a.qkv.bias.detach()[None], out_dtype=adt)[0]
        if pa.adt != qkv.dtype:                                      # attribution: bf16 attention inside an fp32-storage run
            q16 = ops.cast2d(qkv, qkv.shape[0], 3 * C, 3 * C, pa.adt, ldd=3 * C)
            ao16, rawlog, _ = ops.attention(q16, B, N, nH, T, pa)
            ao = ops.cast2d(ao16, ao16.shape[0], C, C, adt, ldd=C)
        else:
            ao, rawlog, _ = ops.attention(qkv, B, N, nH, T, pa)
        XT2 = torch.empty_like(XT)
        ops.linear(ao, ops.pack_linear([a.proj.weight], pe, tag + ('proj',)), C, pe, bias=a.proj.bias.detach()[None],
                   out=XT2, resid=XT)
        # channel attention: queries token_trans(norm1(prompts)), keys norm1(x)^T, windowed (:216-250)
        cq = ops.linear(xn, ops.pack_linear([a.token_trans.weight], ps, tag + ('tt',)), hw, ps,
                        bias=a.token_trans.bias.detach()[None], a_rows=(T, N * C, C), M=B * T, out_dtype=ps.adt)[0]
        rawchan = None
        if self._side_channels_used(i):
            xnc = xn if ps.adt == xn.dtype else ops.cast2d(xn, xn.shape[0], C, C, ps.adt, ldd=C)
            rawchan = ops.chan_logits(cq, xnc, B, T, N, C, grid, (nwin, nwin))
        pr = XT2.view(B, N, C)[:, :T]
        ops.linear(cq, ops.pack_linear([a.token_trans1.weight], ps, tag + ('tt1',)), C, ps,
                   bias=a.token_trans1.bias.detach()[None], out=pr, d_rows=(T, N * C, C), resid=pr, M=B * T)
        xn2, _, _ = ops.layernorm(XT2, blk.norm2.weight.detach(), blk.norm2.bias.detach(), blk.norm2.eps, prec)
        hmid = ops.linear(xn2, ops.pack_linear([blk.mlp.fc1.weight], pe, tag + ('fc1',)), 4 * C, pe,
                          bias=blk.mlp.fc1.bias.detach()[None], act=ACT_GELU, out_dtype=adt)[0]
        XT3 = torch.empty_like(XT)
        ops.linear(hmid, ops.pack_linear([blk.mlp.fc2.weight], pe, tag + ('fc2',)), C, pe,
                   bias=blk.mlp.fc2.bias.detach()[None], out=XT3, resid=XT2)
        return XT3, rawlog, rawchan

    def _block_split(self, blk, tag, XT, B, N, T, grid, nwin, side=True):
        """the block in the x3f mode (inference): the same x3 products, the four big Linears on the LDS-DMA kernel over pre-split
        hi / lo planes (ops.Split) written by the producing kernels (LayerNorm, qkv / fc1 epilogues, attention)."""
        prec, C, nH = self.prec, self.embed_dim, self.num_heads
        hw = grid[0] * grid[1]
        a = blk.attn
        xs, _, _ = ops.layernorm(XT, blk.norm1.weight.detach(), blk.norm1.bias.detach(), blk.norm1.eps, prec, out_dtype="split")
        qkv = ops.linear(xs, ops.pack_linear_split([a.qkv.weight], tag + ('qkv',)), 3 * C, prec, bias=a.qkv.bias.detach()[None],
                         out_dtype="split")[0]
        ao, rawlog, _ = ops.attention(qkv, B, N, nH, T, prec)
        XT2 = torch.empty_like(XT)
        ops.linear(ao, ops.pack_linear_split([a.proj.weight], tag + ('proj',)), C, prec, bias=a.proj.bias.detach()[None], out=XT2, resid=XT)
        # channel attention: its patch rows are read as the planes LayerNorm wrote (no fp32 copy of the normalised tokens), its T prompt rows
        # per image are gathered into a small fp32 matrix
        cq = ops.linear(ops.prompt_rows32(xs, B, N, T, C), ops.pack_linear([a.token_trans.weight], prec, tag + ('tt',)), hw, prec,
                        bias=a.token_trans.bias.detach()[None], M=B * T)[0]
        rawchan = ops.chan_logits(cq, xs, B, T, N, C, grid, (nwin, nwin)) if side else None
        pr = XT2.view(B, N, C)[:, :T]
        ops.linear(cq, ops.pack_linear([a.token_trans1.weight], prec, tag + ('tt1',)), C, prec,
                   bias=a.token_trans1.bias.detach()[None], out=pr, d_rows=(T, N * C, C), resid=pr, M=B * T)
        xs2, _, _ = ops.layernorm(XT2, blk.norm2.weight.detach(), blk.norm2.bias.detach(), blk.norm2.eps, prec, out_dtype="split")
        hmid = ops.linear(xs2, ops.pack_linear_split([blk.mlp.fc1.weight], tag + ('fc1',)), 4 * C, prec,
                          bias=blk.mlp.fc1.bias.detach()[None], act=ACT_GELU, out_dtype="split")[0]
        XT3 = torch.empty_like(XT)
        ops.linear(hmid, ops.pack_linear_split([blk.mlp.fc2.weight], tag + ('fc2',)), C, prec,
                   bias=blk.mlp.fc2.bias.detach()[None], out=XT3, resid=XT2)
        return XT3, rawlog, rawchan

    # ---- cal_task_feature (taskprompter.py:424-487), all tasks at once -----------------------------
    def _decoder_packs(self, il):
        p = self.p
        prec, pf = self._gp('side'), self._gp('fuse')          # fea_decode_* belong to 'side', fea_fuse to 'fuse' (== self.prec unless attributing)
        names = p.TASKS.NAMES
        tar, F = p.embed_dim, p.final_embed_dim
        tarp = ops.pitch(tar)
        dec_w, dec_b = [], []
        for t in names:
            dec_w += [self.fea_decode_spa[il][t][0].weight, self.fea_decode_chan[il][t][0].weight]
            dec_b += [self.fea_decode_spa[il][t][0].bias, self.fea_decode_chan[il][t][0].bias]
        bdec = ops.stack_vec(dec_b, ('decb', il))
        f0 = [self.fea_fuse[il][t][0].weight for t in names]
        # fea_fuse[0] reads torch.cat([spa, chan], 1) (:471): its K = 2*tar columns land at 0 and pitch(tar) of the padded concatenation
        if self._decoder_split():
            Wdec = ops.pack_linear_split(dec_w, ('dec', il))
            W0 = ops.pack_kmap_split(f0, F, 2 * tarp, [(0, 0, tar), (tarp, tar, tar)], ('f0', il))
        else:
            Wdec = ops.pack_linear(dec_w, prec, ('dec', il))
            W0 = ops.pack_kmap(f0, F, 2 * tarp, [(0, 0, tar), (tarp, tar, tar)], pf, ('f0', il))
        b0 = ops.stack_vec([self.fea_fuse[il][t][0].bias for t in names], ('f0b', il))
        fc = [self.fea_fuse[il][t][1].weight for t in names]
        Wc = ops.pack_conv3_split(fc, ('f1', il)) if self._decoder_conv_split() else ops.pack_conv3(fc, pf, ('f1', il))
        bc = ops.stack_vec([self.fea_fuse[il][t][1].bias for t in names], ('f1b', il))
        f4 = [self.fea_fuse[il][t][4].weight for t in names]
        # inference (BatchNorm folded into the conv's epilogue): the conv writes GELU(BN(.)) as planes, so fea_fuse[4] runs on the split-plane
        # kernel too (880 -> 490 us per tap at B = 63, profiles/r05_dec_x3_bench_b_edge.log); in training BatchNorm's apply writes fp32
        W4 = ops.pack_linear_split(f4, ('f4', il)) if self._fuse4_split() else ops.pack_linear(f4, pf, ('f4', il))
        b4 = ops.stack_vec([self.fea_fuse[il][t][4].bias for t in names], ('f4b', il))
        return Wdec, bdec, W0, b0, Wc, bc, W4, b4

    def _decoder_split(self):
        """x3f: fea_decode_* and fea_fuse[0] on the split-plane LDS-DMA kernel — `modulate` writes hi / lo planes, the fea_decode epilogue
        writes the padded concatenation as planes (the register-staged x3 kernel split the fp32 operands of every tile while staging:
        3.2 -> 2.3 ms and 1.2 -> 0.85 ms per tap at B = 63, profiles/r04_dec_x3_bench_g.log).  Needs whole 32-deep K steps."""
        return (self.prec.split and self.gprec is None and ops.split_gemm_ok(self.embed_dim)
                and ops.split_gemm_ok(2 * ops.pitch(self.p.embed_dim)))

    def _decoder_conv_split(self):
        """... and fea_fuse[1] (3x3) on its implicit-GEMM form: fea_fuse[0]'s epilogue writes y0 as planes (channel pitch % 32 == 0)."""
        return self._decoder_split() and ops.split_conv_ok(self.p.final_embed_dim)

    def _fuse4_split(self):
        """... and, in eval mode only, fea_fuse[4] on planes written by the BN-folded conv epilogue (K = pitch(F) a multiple of 32)."""
        return (not self.training) and self._decoder_conv_split() and ops.split_gemm_ok(ops.pitch(self.p.final_embed_dim))

    def _ctr_weights(self, rawlog, il, B, T):
        """[B, T, T] mixing weights: per-head MLP on the prompt<->prompt raw logits (:482-484); identity without ctr."""
        if not self.p.use_ctr:
            return torch.eye(T, dtype=torch.float32, device=rawlog.device)[None].expand(B, T, T).contiguous()
        from . import autograd_path
        mods = [self.ctr_attn_conv[il][t] for t in self.p.TASKS.NAMES]
        return autograd_path.CtrWeightsFn.apply(rawlog, B, T, ('ctrw', il), *[m[0].weight for m in mods], *[m[0].bias for m in mods],
                                                *[m[2].weight for m in mods], *[m[2].bias for m in mods])

    def _bn_fold(self, bns, conv_biases, tag):
        return bn_fold(bns, conv_biases, tag)

    def _bn_train(self, y, bns, C, act):
        """training-mode BatchNorm2d (+act) on [T, rows, ld] pre-activations; updates running stats like nn.BatchNorm2d."""
        from . import bn as bn_mod
        return bn_mod.train_forward(y, C, list(bns), act)[0]

    def _task_features(self, xsrc, xview, rawlog, rawchan, il, B, acc):
        p, prec = self.p, self.prec
        ps, pf, adt = self._gp('side'), self._gp('fuse'), self.prec.adt
        names = p.TASKS.NAMES
        T, C = len(names), self.embed_dim
        h, w = self.resolution
        hw, N = h * w, T + h * w
        tar, F = p.embed_dim, p.final_embed_dim
        tarp, Fp = ops.pitch(tar), ops.pitch(F)
        nwin = int(math.isqrt(self.chan_nheads))
        Wdec, bdec, W0, b0, Wc, bc, W4, b4 = self._decoder_packs(il)
        sp = self._decoder_split()
        mod = ops.modulate(xview, C, N * C, rawlog, rawchan, B, T, N, C, (h, w), (nwin, nwin), prec, split=sp)
        cat = ops.Split.empty((T, B * hw, 2 * tarp), xsrc.device) if sp else torch.empty(T, B * hw, 2 * tarp, dtype=prec.adt, device=xsrc.device)
        ops.linear(mod, Wdec, tar, ps, bias=bdec, out=cat, batch_inner=2, d_z=(B * hw * 2 * tarp, tarp), ldd=2 * tarp,
                   n_store=tarp)
        del mod
        y0 = ops.linear(cat, W0, F, pf, bias=b0, out_dtype="split" if self._decoder_conv_split() else adt)
        del cat
        bns = [self.fea_fuse[il][t][2] for t in names]
        if self.training:
            y1 = ops.conv3x3(y0, Wc, F, F, B, h, w, pf, bias=bc, out_dtype=adt)
            y1 = self._bn_train(y1, bns, F, ACT_GELU)
        else:
            sc, sh = self._bn_fold(bns, [self.fea_fuse[il][t][1].bias for t in names], ('f2', il))
            y1 = ops.conv3x3(y0, Wc, F, F, B, h, w, pf, bias=sh, colscale=sc, act=ACT_GELU, out_dtype="split" if self._fuse4_split() else adt)
        fea = ops.linear(y1, W4, F, pf, bias=b4, out_dtype=adt)
        wmix = self._ctr_weights(rawlog, il, B, T).detach()
        return ops.ctr_mix(fea, wmix, B, F, acc)


def bn_fold(bns, conv_biases, tag):
    """eval BatchNorm folded into the producing conv's epilogue: scale = g/sqrt(v+eps), shift = b + (cb - m)*scale."""
    def build():
        with torch.no_grad():
            sc = torch.stack([bn.weight.detach() * torch.rsqrt(bn.running_var + bn.eps) for bn in bns])
            cb = torch.stack([b.detach() for b in conv_biases])
            sh = torch.stack([bn.bias.detach() - bn.running_mean * s for bn, s in zip(bns, sc)]) + cb * sc
            return sc.contiguous(), sh.contiguous()
    prm = [q for bn in bns for q in (bn.weight, bn.bias, bn.running_mean, bn.running_var)] + list(conv_biases)
    return ops._cached((tag, tuple(id(q) for q in prm)), prm, build)


def _create_task_prompter(variant, pretrained=False, default_cfg=None, **kwargs):
    """taskprompter.py:646-668.  pretrained=True (what the reference's get_backbone passes) loads the variant's ImageNet `.npz` from
    the torch hub cache, where timm's build_model_with_cfg would have downloaded it; never touches the network."""
    kwargs.setdefault('in_chans', 3)
    model = TaskPrompter(**kwargs)
    model.default_cfg = dict(default_cfg or {}, variant=variant)
    if pretrained:
        from .checkpoints import load_cached_pretrained
        load_cached_pretrained(model, variant)
    return model


def taskprompter_vit_large_patch16_384(pretrained=False, **kwargs):
    """ViT-L/16 TaskPrompter (taskprompter.py:671-677)."""
    model_kwargs = dict(select_list=range(6, 24, 6), patch_size=16, embed_dim=1024, depth=24, num_heads=16,
                        chan_nheads=kwargs['p'].chan_nheads, **kwargs)
    return _create_task_prompter('vit_large_patch16_384', pretrained=pretrained, **model_kwargs)


def taskprompter_vit_base_patch16_384(pretrained=False, **kwargs):
    """ViT-B/16 TaskPrompter (taskprompter.py:679-685)."""
    model_kwargs = dict(select_list=range(3, 12, 3), patch_size=16, embed_dim=768, depth=12, num_heads=12,
                        chan_nheads=kwargs['p'].chan_nheads, **kwargs)
    return _create_task_prompter('vit_base_patch16_384', pretrained=pretrained, **model_kwargs)


def _to_rows(x, prec):
    """Reference-layout feature map [B, C, H, W] (any strides; our own channels-last views are copied once) -> NHWC rows
    [1, B*H*W, pitch(C)] in the activation dtype with zero channel padding."""
    B, C, H, W = x.shape
    Cp = ops.pitch(C)
    rows = torch.zeros(1, B * H * W, Cp, dtype=prec.adt, device=x.device)
    rows.view(B, H, W, Cp)[..., :C] = x.permute(0, 2, 3, 1)
    return rows


class _HeadBase(nn.Module):
    """A prediction head is usable on its own, like the reference's (taskprompter.py:688-715): `head(x)` with x [B, C, H, W] returns
    fp32 [B, n_out, H', W'] computed by the same HIP kernels the wrapper batches over tasks (differentiable in training)."""
    prec = None

    def forward(self, x):
        prec = self.prec or ops.Prec(ops.DEFAULT_PREC)
        B, _, H, W = x.shape
        rows = _to_rows(x, prec)
        kind = 'conv' if isinstance(self, ConvHead) else 'deconv'
        return run_heads(kind, [self], rows, B, H, W, None, prec, self.training)[0]


class ConvHead(_HeadBase):
    """taskprompter.py:688-698 (parameter holder; executed task-batched by TaskPrompterWrapper)."""

    def __init__(self, in_channels, num_classes):
        super().__init__()
        self.mt_proj = nn.Sequential(nn.Conv2d(in_channels, in_channels, 3, padding=1), BatchNorm2d(in_channels), nn.GELU())
        trunc_normal_(self.mt_proj[0].weight, std=0.02)
        self.linear_pred = nn.Conv2d(in_channels, num_classes, kernel_size=1)


class DEConvHead(_HeadBase):
    """taskprompter.py:700-715."""

    def __init__(self, in_channels, num_classes):
        super().__init__()
        self.mt_proj = nn.Sequential(
            nn.ConvTranspose2d(in_channels, in_channels // 2, 2, stride=2, padding=0), BatchNorm2d(in_channels // 2), nn.GELU(),
            nn.Conv2d(in_channels // 2, in_channels // 2, 3, padding=1), BatchNorm2d(in_channels // 2), nn.GELU())
        self.linear_pred = nn.Conv2d(in_channels // 2, num_classes, kernel_size=1)
        trunc_normal_(self.mt_proj[0].weight, std=0.02)
        trunc_normal_(self.mt_proj[3].weight, std=0.02)
        trunc_normal_(self.linear_pred.weight, std=0.02)


def run_heads(kind, heads, fea, B, h4, w4, target, prec, training, lowres=False):
    """The per-task prediction heads of one kind on the task stack fea [Z, B*h4*w4, pitch(F)] -> list of fp32 NCHW predictions resized to
    `target` (None: the head's native resolution).  ConvHead: ONE task-batched 3x3 conv + BN + GELU, then the 1x1s (taskprompter.py:
    688-698); DEConvHead: ConvT 2x2 s2 as a pixel-shuffle GEMM + BN + GELU + 3x3 + BN + GELU + 1x1 (:700-715).
    lowres (ConvHeads only): fea is the backbone's result BEFORE its x4 resize, [Z, B*(h4/4)*(w4/4), pitch(F)], and the resize is fused
    into the 3x3 conv in its taps-first form (ops.upconv3x3) — the upsampled features are never materialised."""
    if torch.is_grad_enabled() and (fea.requires_grad or any(q.requires_grad for hd in heads for q in hd.parameters())):
        from . import autograd_path
        return autograd_path.heads_forward(kind, heads, fea, B, h4, w4, target, prec, training, lowres)
    from . import bn as bn_mod
    F = heads[0].mt_proj[0].weight.shape[0]
    outs = []
    if kind == 'conv':
        tgt = target or (h4, w4)
        conv_w = [hd.mt_proj[0].weight for hd in heads]
        bns = [hd.mt_proj[1] for hd in heads]
        if lowres:
            if fea.dtype != prec.adt:
                fea = ops.cast_rows(fea.reshape(-1, fea.shape[-1]), prec.adt).view(fea.shape)
            sp9 = prec.split and ops.split_gemm_ok(ops.pitch(conv_w[0].shape[1]))
            W9 = ops.pack_upconv9_split(conv_w, 'hc9') if sp9 else ops.pack_upconv9(conv_w, prec, 'hc9')

            def conv(**epi):
                return ops.upconv3x3(fea, W9, F, B, h4 // 4, w4 // 4, prec, **epi)
        else:
            Wc = ops.pack_conv3(conv_w, prec, 'hc')

            def conv(**epi):
                return ops.conv3x3(fea, Wc, F, F, B, h4, w4, prec, **epi)
        if training:
            y = conv(bias=ops.stack_vec([hd.mt_proj[0].bias for hd in heads], 'hcb'))
            y = bn_mod.train_forward(y, F, bns, ACT_GELU)[0]
        else:
            sc, sh = bn_fold(bns, [hd.mt_proj[0].bias for hd in heads], 'hbn')
            y = conv(bias=sh, colscale=sc, act=ACT_GELU)
        for i, hd in enumerate(heads):
            n_out = hd.linear_pred.weight.shape[0]
            pred = ops.linear(y[i], ops.pack_linear([hd.linear_pred.weight], prec, 'hp'), n_out, prec,
                              bias=hd.linear_pred.bias.detach()[None], out_dtype=torch.float32)
            outs.append(ops.bilinear(pred, B, n_out, h4, w4, tgt[0], tgt[1], torch.float32, nchw=True))
        return outs
    F2 = F // 2
    tgt = target or (2 * h4, 2 * w4)
    for i, hd in enumerate(heads):
        wt = hd.mt_proj[0].weight                                   # [F, F2, 2, 2] -> rows n = (dy*2+dx)*F2 + co
        Wd = ops._cached(('hd0', prec.name, id(wt)), [wt],
                         lambda wt=wt: ops.pack_matrix(wt.detach().permute(2, 3, 1, 0).reshape(4 * F2, F), prec)[None])
        b4v = ops._cached(('hd0b', id(hd.mt_proj[0].bias)), [hd.mt_proj[0].bias],
                          lambda hd=hd: hd.mt_proj[0].bias.detach().repeat(4).contiguous())
        y = ops.deconv2x2(fea[i], Wd, F2, F, B, h4, w4, prec, bias4=b4v)[None]      # [1, B*2h4*2w4, pitch(F2)]
        Wc = ops.pack_conv3([hd.mt_proj[3].weight], prec, 'hd3')
        bc = hd.mt_proj[3].bias.detach()[None].contiguous()
        if training:
            y = bn_mod.train_forward(y, F2, [hd.mt_proj[1]], ACT_GELU)[0]
            y = ops.conv3x3(y, Wc, F2, F2, B, 2 * h4, 2 * w4, prec, bias=bc)
            y = bn_mod.train_forward(y, F2, [hd.mt_proj[4]], ACT_GELU)[0]
        else:
            bn1 = hd.mt_proj[1]
            y = ops.bn_apply(y, F2, bn1.running_mean, torch.rsqrt(bn1.running_var + bn1.eps), bn1.weight.detach(),
                             bn1.bias.detach(), ACT_GELU)
            sc, sh = bn_fold([hd.mt_proj[4]], [hd.mt_proj[3].bias], 'hd4')
            y = ops.conv3x3(y, Wc, F2, F2, B, 2 * h4, 2 * w4, prec, bias=sh, colscale=sc, act=ACT_GELU)
        n_out = hd.linear_pred.weight.shape[0]
        pred = ops.linear(y[0], ops.pack_linear([hd.linear_pred.weight], prec, 'hp'), n_out, prec,
                          bias=hd.linear_pred.bias.detach()[None], out_dtype=torch.float32)
        outs.append(ops.bilinear(pred, B, n_out, 2 * h4, 2 * w4, tgt[0], tgt[1], torch.float32, nchw=True))
    return outs


class TaskPrompterWrapper(nn.Module):
    """taskprompter_wrapper.py:9-40: backbone -> per-task head -> bilinear resize to the input size (or `dd_label_map_size`).
    ConvHead / DEConvHead tasks run task-batched on the HIP kernels; any other head module (a user's own, or the reference's '3ddet'
    head) is called as `heads[task](feature)` on the reference-layout feature map, and a '3ddet' output is passed through un-resized
    (taskprompter_wrapper.py:35-38)."""

    fuse_upsample = True        # False: materialise the x4-upsampled features and run the 3x3 conv on them (A/B + parity tests)

    def __init__(self, p, backbone, heads):
        super().__init__()
        self.tasks = p.TASKS.NAMES
        self.backbone = backbone
        self.heads = heads
        self.target_size = p.dd_label_map_size if 'dd_label_map_size' in p.keys() else None
        for hd in heads.values():
            if isinstance(hd, _HeadBase):
                hd.prec = backbone.prec

    def forward(self, x):
        img_size = tuple(x.shape[-2:])
        target = tuple(self.target_size) if self.target_size is not None else img_size
        bb = self.backbone
        B = x.shape[0]
        # the grid the heads run on: the ViT backbone's x4-upsampled patch grid, or what the backbone says (Swin: first level x2)
        h4, w4 = getattr(bb, 'feature_hw', None) or (bb.resolution[0] * 4, bb.resolution[1] * 4)
        F = bb.p.final_embed_dim
        out = {}
        groups = {'conv': [], 'deconv': [], 'other': []}
        for i, t in enumerate(self.tasks):
            hd = self.heads[t]
            groups['conv' if isinstance(hd, ConvHead) else ('deconv' if isinstance(hd, DEConvHead) else 'other')].append(i)
        # ConvHeads take the backbone's h x w sums and fuse its x4 resize into their 3x3 conv (taskprompter.py:420 -> :692, taps first);
        # the upsampled stack [T, B*4h*4w, Fp] is only built if some other head needs it
        fuse = self.fuse_upsample and bool(groups['conv']) and getattr(bb, 'fusable_upsample', True)
        lo = bb.forward_nhwc(x, upsample=not fuse)
        fea = None if fuse else lo
        for kind in ('conv', 'deconv'):
            idx = groups[kind]
            if not idx:
                continue
            lowres = fuse and kind == 'conv'
            if not lowres and fea is None:
                fea = bb.upsample4(lo, B)                                   # [T, B*4h*4w, Fp]
            src = lo if lowres else fea
            sub = src if len(idx) == len(self.tasks) else src[idx]
            hprec = bb._gp('heads') if hasattr(bb, '_gp') else bb.prec
            preds = run_heads(kind, [self.heads[self.tasks[i]] for i in idx], sub, B, h4, w4, target, hprec, self.training, lowres)
            for i, pr in zip(idx, preds):
                out[self.tasks[i]] = pr
        if groups['other'] and fea is None:
            fea = bb.upsample4(lo, B)
        for i in groups['other']:
            t = self.tasks[i]
            y = self.heads[t](fea[i].view(B, h4, w4, -1)[..., :F].permute(0, 3, 1, 2).float())
            out[t] = y if t == '3ddet' else torch.nn.functional.interpolate(y, target, mode=INTERPOLATE_MODE)
        return {t: out[t] for t in self.tasks}
