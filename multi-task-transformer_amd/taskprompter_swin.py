"""MI355X-native TaskPrompter on a Swin backbone (TaskPrompter/models/transformers/taskprompter_swin.py): the
reference's constructor arguments and state_dict layout (so `load_state_dict(strict=True)` takes its checkpoints and the module drops
into TaskPrompterWrapper / get_model), executed as a fused schedule on libmtt_hip.so.

Status: the forward (inference) path below is the fused one; with gradients enabled `forward_nhwc` builds the autograd graph of
swin_autograd.py (first, unoptimised training path: HIP kernels for the heavy operators, torch glue for residual adds and row
concatenations, DropPath 0 only).

Schedule of a SwinTransformerBlock (taskprompter_swin.py:324-414), per image a token buffer XT [T + H*W, C] fp32 with the T task
prompts first (like the ViT variant):
  LayerNorm(norm1) over all rows -> mtt_gather_rows builds the window token matrix [nW, T + ws^2, C] (cyclic shift, zero padding and
  the prompts joined to every window in ONE gather through a precomputed index table) -> qkv GEMM -> mtt_winattn_fwd (relative-position
  bias, shift mask, softmax, PV on MFMA; the prompt rows' raw logits go straight into image-layout maps: un-shifted, un-padded)
  -> inverse gather back to image order (the prompt rows averaged over the windows first: the mean commutes with the proj Linear)
  -> proj GEMM -> residual add; channel attention: chan_kv as a GEMM with the pixel axis as the reduction (no transposed copy of the
  feature map), mtt_chanattn_fwd (logits over the channels, softmax, value mix); prompt update GEMMs; MLP over prompts + pixels at once.
PatchMerging (:439-472): 4 row gathers -> LayerNorm -> reduction GEMM; the attention maps through mtt_conv3s2_nchw; channel logits and
prompts through small GEMMs.  cal_task_feature (:715-777): the 1x1 decoders and fea_fuse[0] run BEFORE the x2 bilinear resize (1x1
convolutions commute with it: 4x fewer MACs and bytes), then 3x3 + BN + GELU + 3x3.  Multi-scale fusion: bilinear kernels accumulate
into the first level's grid, then the 3x3 multi_scale_fuse conv; heads as in the ViT variant.
"""
import math

import torch
import torch.nn as nn

from . import ops
from ._lib import ACT_GELU, F32, OP_K, OP_R, dtype_code
from .taskprompter import BatchNorm2d, Mlp, _init_vit_weights, _prec_of, bn_fold, trunc_normal_


# ---- geometry (host side, cached): index tables of the window partition / reverse / merging ----------------------------------------
def block_geometry(res, window, shift_flag):
    """taskprompter_swin.py:243-247, 262-272 -> (ws, shift, Hp, Wp)."""
    H, W = res
    ws, shift = window, (window // 2 if shift_flag else 0)
    if min(res) <= ws:
        ws, shift = min(res), 0
    return ws, shift, H + (ws - H % ws) % ws, W + (ws - W % ws) % ws


def relative_position_index(ws):
    """taskprompter_swin.py:147-157."""
    c = torch.stack(torch.meshgrid(torch.arange(ws), torch.arange(ws), indexing="ij")).flatten(1)
    rel = (c[:, :, None] - c[:, None, :]).permute(1, 2, 0) + (ws - 1)
    return rel[..., 0] * (2 * ws - 1) + rel[..., 1]


def shift_attn_mask(Hp, Wp, ws, shift):
    """taskprompter_swin.py:281-300: [nW, ws*ws, ws*ws] of 0 / -100, or None."""
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
    return torch.where(diff != 0, torch.tensor(-100.0), torch.tensor(0.0)).contiguous()


_geo_cache = {}


def window_tables(res, ws, shift, Hp, Wp, T, device):
    """part [nW*(T+ws^2)]: source row (in the image's T + H*W token rows) of every window token (-1 = zero padding);
    pix [nW, ws^2]: image pixel of every window position (-1 = padding); rev [H*W]: window-layout row of every image pixel."""
    key = (tuple(res), ws, shift, Hp, Wp, T, str(device))
    hit = _geo_cache.get(key)
    if hit is not None:
        return hit
    H, W = res
    nWh, nWw = Hp // ws, Wp // ws
    ys = torch.arange(Hp).view(nWh, ws)[:, None, :, None]            # shifted-map coordinates of (window, position)
    xs = torch.arange(Wp).view(nWw, ws)[None, :, None, :]
    y = (ys + shift) % Hp                                             # shifted[ys][xs] = padded[(ys + shift) % Hp][(xs + shift) % Wp]
    x = (xs + shift) % Wp
    ok = (y < H) & (x < W)
    pix = torch.where(ok, y * W + x, torch.full_like(y * W + x, -1)).reshape(nWh * nWw, ws * ws)
    Nw = T + ws * ws
    part = torch.empty(nWh * nWw, Nw, dtype=torch.int64)
    part[:, :T] = torch.arange(T)[None]
    part[:, T:] = torch.where(pix >= 0, pix + T, pix)
    yy = torch.arange(H)[:, None]
    xx = torch.arange(W)[None, :]
    ysr, xsr = (yy - shift) % Hp, (xx - shift) % Wp
    rev = ((ysr // ws) * nWw + xsr // ws) * Nw + T + (ysr % ws) * ws + xsr % ws
    out = (part.reshape(-1).to(torch.int32).to(device), pix.to(torch.int32).contiguous().to(device), rev.reshape(-1).to(torch.int32).to(device))
    _geo_cache[key] = out
    return out


def merge_tables(res, T, device):
    """PatchMerging's x0..x3 (taskprompter_swin.py:450-454): 4 source-row tables [H/2 * W/2]."""
    key = ('merge', tuple(res), T, str(device))
    hit = _geo_cache.get(key)
    if hit is not None:
        return hit
    H, W = res
    yy = torch.arange(H // 2)[:, None] * 2
    xx = torch.arange(W // 2)[None, :] * 2
    out = [(T + (yy + dy) * W + (xx + dx)).reshape(-1).to(torch.int32).to(device) for dy, dx in ((0, 0), (1, 0), (0, 1), (1, 1))]
    _geo_cache[key] = out
    return out


def _gather(src, dst, idx, rows, C, ld_src, ld_dst, B, src_bs, dst_bs):
    ops.call("gather_rows", src=src, dst=dst, idx=idx, rows=rows, C=C, ld_src=ld_src, ld_dst=ld_dst, src_dtype=dtype_code(src),
             dst_dtype=dtype_code(dst), B=B, src_bs=src_bs, dst_bs=dst_bs, idx_bs=0)


# ---- parameter holders with the reference's names -----------------------------------------------------------------------------------
class PatchEmbed(nn.Module):
    def __init__(self, img_size, patch_size, in_chans, embed_dim, norm_layer):
        super().__init__()
        self.img_size = tuple(img_size)
        self.patch_size = (patch_size, patch_size)
        self.grid_size = (img_size[0] // patch_size, img_size[1] // patch_size)
        self.num_patches = self.grid_size[0] * self.grid_size[1]
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size)
        self.norm = norm_layer(embed_dim) if norm_layer else nn.Identity()


class WindowAttention(nn.Module):
    """taskprompter_swin.py:120-166."""

    def __init__(self, dim, window_size, num_heads, qkv_bias=True):
        super().__init__()
        self.dim, self.window_size, self.num_heads = dim, window_size, num_heads
        self.relative_position_bias_table = nn.Parameter(torch.zeros((2 * window_size - 1) ** 2, num_heads))
        self.register_buffer("relative_position_index", relative_position_index(window_size))
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim)
        trunc_normal_(self.relative_position_bias_table, std=.02)


class SwinTransformerBlock(nn.Module):
    """taskprompter_swin.py:213-322."""

    def __init__(self, last_block, p, dim, input_resolution, num_heads, window_size, shift_flag, mlp_ratio, qkv_bias, drop_path):
        super().__init__()
        self.last_block = last_block
        self.dim, self.input_resolution, self.num_heads = dim, tuple(input_resolution), num_heads
        self.window_size, self.shift_size, Hp, Wp = block_geometry(self.input_resolution, window_size, shift_flag)
        self.padded = (Hp, Wp)
        self.norm1 = nn.LayerNorm(dim)
        self.attn = WindowAttention(dim, self.window_size, num_heads, qkv_bias)
        self.drop_path_rate = float(drop_path)
        self.norm2 = nn.LayerNorm(dim)
        self.mlp = Mlp(dim, int(dim * mlp_ratio))
        self.register_buffer("attn_mask", shift_attn_mask(Hp, Wp, self.window_size, self.shift_size))
        pixel_no = int(input_resolution[0] * input_resolution[1])
        ce = p.chan_embed_dim
        self.chan_q = nn.Linear(ce, ce, bias=qkv_bias)
        self.chan_kv = nn.Linear(pixel_no, ce * 2, bias=qkv_bias)
        self.token_trans = nn.Linear(dim, ce)
        if not last_block:
            self.chan_proj = nn.Linear(ce, ce)
            self.token_trans1 = nn.Linear(ce, dim)


class PatchMerging(nn.Module):
    """taskprompter_swin.py:417-437."""

    def __init__(self, p, num_heads, input_resolution, dim):
        super().__init__()
        self.input_resolution, self.dim = tuple(input_resolution), dim
        self.reduction = nn.Linear(4 * dim, 2 * dim, bias=False)
        self.norm = nn.LayerNorm(4 * dim)
        task_no = len(p.TASKS.NAMES)
        self.process_chan_attn = nn.Linear(dim, 2 * dim, bias=False)
        self.task_prompts_up = nn.Linear(dim, 2 * dim, bias=False)
        self.spa_attn_ds = nn.Conv2d(num_heads * task_no, num_heads * task_no, kernel_size=3, padding=1, stride=2)


class BasicLayer(nn.Module):
    """taskprompter_swin.py:487-543."""

    def __init__(self, last_layer, p, dim, input_resolution, depth, num_heads, window_size, mlp_ratio, qkv_bias, drop_path, downsample):
        super().__init__()
        self.dim, self.input_resolution, self.depth = dim, tuple(input_resolution), depth
        self.blocks = nn.ModuleList([
            SwinTransformerBlock(last_layer and i == depth - 1, p, dim, input_resolution, num_heads, window_size, i % 2 == 1, mlp_ratio,
                                 qkv_bias, drop_path[i]) for i in range(depth)])
        self.downsample = PatchMerging(p, num_heads, input_resolution, dim) if downsample else None


class TaskPrompterSwin(nn.Module):
    """TaskPrompter built upon Swin Transformer (taskprompter_swin.py:546-777)."""

    fusable_upsample = False     # TaskPrompterWrapper: this backbone does not end in a x4 resize that ConvHeads could absorb

    def __init__(self, p=None, img_size=224, patch_size=4, in_chans=3, embed_dim=96, depths=(2, 2, 6, 2), num_heads=(3, 6, 12, 24),
                 window_size=7, mlp_ratio=4., qkv_bias=True, drop_rate=0., attn_drop_rate=0., drop_path_rate=0.1, norm_layer=nn.LayerNorm,
                 ape=False, patch_norm=True, use_checkpoint=False, weight_init='', **kwargs):
        super().__init__()
        assert in_chans == 3 and patch_size in (2, 4, 8) and not ape and drop_rate == 0. and attn_drop_rate == 0.
        assert all(embed_dim * 2 ** i // nh == 32 for i, nh in enumerate(num_heads)), "HIP window attention: head_dim 32 (every Swin size)"
        assert p.prompt_len == 1, "as every reference config (the ViT variant asserts it for its cross-task step)"
        self.p = p
        self.prec = _prec_of(p)
        self.num_layers = len(depths)
        self.embed_dim = embed_dim
        self.num_features = int(embed_dim * 2 ** (self.num_layers - 1))
        self.img_ds_ratio = p.img_ds_ratio
        self.resolution = [[int(s[0] * self.img_ds_ratio), int(s[1] * self.img_ds_ratio)] for s in p.ori_spatial_dim]     # :594-596
        self.full_img_size = tuple(img_size)
        img_size = [int(v * self.img_ds_ratio) for v in img_size]
        self.patch_embed = PatchEmbed(img_size, patch_size, in_chans, embed_dim, norm_layer if patch_norm else None)
        self.patch_grid = self.patch_embed.grid_size
        dpr = [x.item() for x in torch.linspace(0, drop_path_rate, sum(depths))]
        self.task_no = len(p.TASKS.NAMES)
        self.all_tasks = p.TASKS.NAMES
        self.prompt_len = p.prompt_len
        self.prompts_len = self.task_no * p.prompt_len
        p.prompts_len = self.prompts_len
        self.task_prompts = nn.Parameter(torch.ones(self.prompts_len, embed_dim))
        trunc_normal_(self.task_prompts, mean=1., std=1.)
        self.fea_fuse, self.fea_decode_spa, self.fea_decode_chan = nn.ModuleList(), nn.ModuleList(), nn.ModuleList()
        for il in range(self.num_layers):
            cur, tar, fin = p.backbone_channels[il], p.level_embed_dim, p.final_embed_dim
            self.fea_fuse.append(nn.ModuleDict())
            self.fea_decode_spa.append(nn.ModuleDict())
            self.fea_decode_chan.append(nn.ModuleDict())
            for task in p.TASKS.NAMES:
                self.fea_fuse[il][task] = nn.Sequential(nn.Conv2d(tar * 2, fin, kernel_size=1), nn.Conv2d(fin, fin, kernel_size=3, padding=1),
                                                        BatchNorm2d(fin), nn.GELU(), nn.Conv2d(fin, fin, kernel_size=3, padding=1))
                self.fea_decode_spa[il][task] = nn.Sequential(nn.Conv2d(cur, tar, kernel_size=1, padding=0))
                self.fea_decode_chan[il][task] = nn.Sequential(nn.Conv2d(cur, tar, kernel_size=1, padding=0))
        self.multi_scale_fuse = nn.ModuleDict({t: nn.Conv2d(p.final_embed_dim, p.final_embed_dim, kernel_size=3, padding=1)
                                               for t in p.TASKS.NAMES if t != '3ddet'})
        layers = []
        for il in range(self.num_layers):
            layers.append(BasicLayer(il == self.num_layers - 1, p, int(embed_dim * 2 ** il),
                                     (self.patch_grid[0] // 2 ** il, self.patch_grid[1] // 2 ** il), depths[il], num_heads[il], window_size,
                                     mlp_ratio, qkv_bias, dpr[sum(depths[:il]):sum(depths[:il + 1])], il < self.num_layers - 1))
        self.layers = nn.Sequential(*layers)
        self.norm = norm_layer(self.num_features)
        self.apply(_init_vit_weights)
        # the task features leave at the first level's x2-upsampled grid (the wrapper runs the heads there)
        self.feature_hw = (2 * self.resolution[0][0], 2 * self.resolution[0][1])

    # ---- API used by TaskPrompterWrapper -------------------------------------------------------------------------------------------
    def forward(self, x):
        """-> ({task: [B, F, h, w]}, info) as taskprompter_swin.py:664-713."""
        fea = self.forward_nhwc(x)
        B, F = x.shape[0], self.p.final_embed_dim
        h, w = self.feature_hw
        return {t: fea[i].view(B, h, w, -1)[..., :F].permute(0, 3, 1, 2) for i, t in enumerate(self.all_tasks)}, {}

    def upsample4(self, acc, B):
        raise RuntimeError("the Swin backbone returns its features at their final resolution")

    def forward_nhwc(self, img, upsample=True):
        if torch.is_grad_enabled() and any(q.requires_grad for q in self.parameters()):
            from . import swin_autograd
            return swin_autograd.backbone_forward(self, img)
        return self._forward_nograd(img)

    # ---- forward ---------------------------------------------------------------------------------------------------------------------
    def _decoder_split(self, C):
        """x3f: a level's fea_decode_* and fea_fuse[0] on the split-plane LDS-DMA kernel (`modulate` and the fea_decode epilogue write
        hi / lo planes) — whole 32-deep K steps over the level's C channels and over the padded concatenation (TaskPrompter._decoder_split)."""
        return self.prec.split and ops.split_gemm_ok(C) and ops.split_gemm_ok(2 * ops.pitch(self.p.level_embed_dim))

    SPLIT_MIN_ROWS = 2048

    def _lin(self, x, layer, tag, **kw):
        """x @ layer.weight^T (+ bias) through mtt_gemm; kw as ops.linear.  x3f: a large fp32 operand is split into hi / lo planes by one
        pass and the product runs on the split-plane LDS-DMA kernel (pre-split weight planes) instead of the register-staged x3 one."""
        N, K = layer.weight.shape[0], layer.weight.shape[1]
        bias = layer.bias.detach()[None] if layer.bias is not None else None
        if (self.prec.split and torch.is_tensor(x) and x.dtype == torch.float32 and x.dim() == 2 and x.is_contiguous() and x.shape[1] == K
                and kw.get("a_rows") is None and x.shape[0] >= self.SPLIT_MIN_ROWS and ops.split_gemm_ok(K)):
            return ops.linear(ops.split_cast(x), ops.pack_linear_split([layer.weight], tag), N, self.prec, bias=bias, **kw)
        return ops.linear(x, ops.pack_linear([layer.weight], self.prec, tag), N, self.prec, bias=bias, **kw)

    def _forward_nograd(self, img):
        p, prec = self.p, self.prec
        dev = img.device
        B = img.shape[0]
        assert tuple(img.shape[-2:]) == self.full_img_size, "input size must equal img_size"
        img = img.float().contiguous()
        if self.img_ds_ratio != 1:                                          # :666-667
            Hs, Ws = self.patch_embed.img_size
            small = torch.empty(B, 3, Hs, Ws, dtype=torch.float32, device=dev)
            ops.call("resize_nchw", args=[img, small, B * 3, img.shape[-2], img.shape[-1], Hs, Ws])
            img = small
        T = self.prompts_len
        ps = self.patch_embed.patch_size[0]
        gh, gw = self.patch_grid
        C = self.embed_dim
        N = T + gh * gw
        # ---- patch embed (+ patch_norm) into the token buffer, prompts first -------------------------------------------------------
        Kp = ops.pitch(3 * ps * ps)
        cols = torch.empty(B * gh * gw, Kp, dtype=prec.adt, device=dev)
        ops.call("patchify", args=[img, cols, B, img.shape[-2], img.shape[-1], ps, Kp, dtype_code(cols)])
        XT = torch.empty(B * N, C, dtype=torch.float32, device=dev)
        if isinstance(self.patch_embed.norm, nn.LayerNorm):
            pe = torch.zeros(B * N, C, dtype=torch.float32, device=dev)         # prompt rows: zeros (normalised, then overwritten)
            self._lin(cols, self.patch_embed.proj, 'pe', out=pe.view(B, N, C)[:, T:], d_rows=(gh * gw, N * C, C), M=B * gh * gw)
            ops.call("layernorm_fwd", x=pe, y=XT, gamma=self.patch_embed.norm.weight.detach(), beta=self.patch_embed.norm.bias.detach(),
                     mean=None, rstd=None, rows=B * N, C=C, ldx=C, ldy=C, y_dtype=F32, eps=self.patch_embed.norm.eps)
        else:
            self._lin(cols, self.patch_embed.proj, 'pe', out=XT.view(B, N, C)[:, T:], d_rows=(gh * gw, N * C, C), M=B * gh * gw)
        XT.view(B, N, C)[:, :T] = self.task_prompts.detach()

        fea_levels = []
        rawlog = rawchan = None
        nl = self.num_layers
        for il, layer in enumerate(self.layers):
            res = layer.input_resolution
            for ib, blk in enumerate(layer.blocks):
                XT, rawlog, rawchan = self._block(blk, (il, ib), XT, B, T, res)
            if layer.downsample is not None:
                XT, rawlog, rawchan = self._merge(layer.downsample, il, XT, rawlog, rawchan, B, T, res, layer.blocks[0].num_heads)
                C2 = 2 * layer.dim
                r2 = (res[0] // 2, res[1] // 2)
                fea_levels.append(self._task_features(XT, XT.view(B, T + r2[0] * r2[1], C2)[:, T:], rawlog, rawchan, il, B, r2, C2,
                                                      C2 // layer.blocks[0].num_heads))
        res = self.layers[-1].input_resolution
        Cl = self.layers[-1].dim
        xf, _, _ = ops.layernorm(XT, self.norm.weight.detach(), self.norm.bias.detach(), self.norm.eps, prec, out_dtype=torch.float32)
        Nl = T + res[0] * res[1]
        fea_levels.append(self._task_features(xf, xf.view(B, Nl, Cl)[:, T:], rawlog, rawchan, nl - 1, B, res, Cl,
                                              Cl // self.layers[-1].blocks[0].num_heads))
        # ---- multi-scale fusion (:699-709): every level resized to the first one's grid and summed, then a 3x3 conv ----------------
        h0, w0 = self.feature_hw
        Fp = fea_levels[0].shape[-1]
        Tn = len(self.all_tasks)
        acc = torch.empty(Tn, B * h0 * w0, Fp, dtype=torch.float32, device=dev)
        for i, f in enumerate(fea_levels):
            hi, wi = 2 * self.resolution[i][0], 2 * self.resolution[i][1]
            ops.call("bilinear_fwd", **{"in": f}, out=acc, B=Tn * B, C=Fp, Hin=hi, Win=wi, Hout=h0, Wout=w0, ld_in=Fp, ld_out=Fp,
                     in_dtype=dtype_code(f), out_dtype=F32, out_nchw=0, accumulate=1 if i else 0)
        F = p.final_embed_dim
        names = [t for t in self.all_tasks]
        Wm = ops.pack_conv3([self.multi_scale_fuse[t].weight for t in names], prec, 'msf')
        bm = ops.stack_vec([self.multi_scale_fuse[t].bias for t in names], 'msfb')
        accq = acc if prec.adt == torch.float32 else ops.cast2d(acc.view(-1, Fp), Tn * B * h0 * w0, Fp, Fp, prec.adt, ldd=Fp).view(Tn, -1, Fp)
        return ops.conv3x3(accq, Wm, F, F, B, h0, w0, prec, bias=bm)

    def _block(self, blk, tag, XT, B, T, res):
        prec = self.prec
        dev = XT.device
        H, W = res
        C, nH, ws, shift = blk.dim, blk.num_heads, blk.window_size, blk.shift_size
        Hp, Wp = blk.padded
        N, ws2 = T + H * W, ws * ws
        Nw, nW = T + ws2, (Hp // ws) * (Wp // ws)
        part, pix, rev = window_tables(res, ws, shift, Hp, Wp, T, dev)
        tag = ('sw',) + tag
        a = blk.attn
        # norm1 over prompts + pixels; the raw prompts feed the channel branch (:331-332)
        xn, _, _ = ops.layernorm(XT, blk.norm1.weight.detach(), blk.norm1.bias.detach(), blk.norm1.eps, prec)
        chan_p = self._lin(XT, blk.token_trans, tag + ('tt',), a_rows=(T, N * C, C), M=B * T, out_dtype=torch.float32)[0]   # [B*T, ce]
        # window tokens: shift + pad + prompts in one gather
        wtok = torch.empty(B * nW * Nw, C, dtype=prec.adt, device=dev)
        _gather(xn, wtok, part, nW * Nw, C, C, C, B, N * C, nW * Nw * C)
        qkv = self._lin(wtok, a.qkv, tag + ('qkv',))[0]
        bias = ops._cached((tag, 'rpb'), [a.relative_position_bias_table],
                           lambda: a.relative_position_bias_table.detach()[a.relative_position_index.view(-1)]
                           .view(ws2, ws2, nH).permute(2, 0, 1).contiguous().float())
        ao = torch.empty(B * nW * Nw, C, dtype=prec.adt, device=dev)
        rawlog = torch.zeros(B, nH, T, N, dtype=torch.float32, device=dev)
        ops.call("winattn_fwd", qkv=qkv, out=ao, rawmap=rawlog, bias=bias, mask=blk.attn_mask, pix=pix, nwin=B * nW, nW=nW, nH=nH, T=T,
                 ws2=ws2, dtype=dtype_code(qkv), scale=32 ** -0.5, map_ld=N, map_off=T, mfma=1 if qkv.dtype == torch.float32 else 0)
        # back to image order; prompt rows = mean over the windows (:208; the mean commutes with proj)
        ao_img = torch.empty(B * N, C, dtype=prec.adt, device=dev)
        _gather(ao, ao_img.view(B, N, C)[:, T:], rev, H * W, C, C, C, B, nW * Nw * C, N * C)
        ao_img.view(B, N, C)[:, :T] = ao.view(B, nW, Nw, C)[:, :, :T].float().mean(1).to(prec.adt)
        po = self._lin(ao_img, a.proj, tag + ('proj',))[0]                                               # [B*N, C] activation dtype
        XT2 = XT                                                                 # the residual stream is updated in place
        ops.call("add_rows", args=[po, XT2, B * N, C, C, C, dtype_code(po), 1.0])
        # channel attention (:391-409): kv^T = W_kv [2ce, HW] x x_attn [HW, C] per image — pixels are the reduction axis
        ce = self.p.chan_embed_dim
        nwin = int(math.isqrt(self.p.chan_nheads))
        q = self._lin(chan_p, blk.chan_q, tag + ('cq',), out_dtype=torch.float32)[0]                     # [B*T, ce]
        Wkv = ops.pack_linear([blk.chan_kv.weight], prec, tag + ('ckv',))                                # [1, 2ce, pitch(HW)]
        Cp = ops.pitch(C)
        pov = po.view(B, N, C)[:, T:]
        K = H * W
        Ks = _split_k(K)                                                        # the reduction axis is the pixel count (73 728 at the first
        S = K // Ks                                                             # Swin-B stage) and M x N is tiny: split it over workgroups
        slabs = torch.empty(B, S, 2 * ce, Cp, dtype=torch.float32, device=dev)
        ops.call("gemm", A=Wkv, B=pov, D=slabs, M=2 * ce, N=C, K=Ks, a_op=OP_K, b_op=OP_R, a_dtype=dtype_code(Wkv), b_dtype=dtype_code(po),
                 d_dtype=F32, prec=prec.code, lda=Wkv.shape[-1], ldb=C, ldd=Cp, batch=B * S, batch_inner=S, a_zo=0, a_zi=Ks, b_zo=N * C,
                 b_zi=Ks * C, d_zo=S * 2 * ce * Cp, d_zi=2 * ce * Cp, alpha=1.0, n_store=Cp)
        kvT = slabs.sum(1) if S > 1 else slabs.view(B, 2 * ce, Cp)
        rawchan = torch.empty(B, T, nwin * nwin, C, dtype=torch.float32, device=dev)
        cx = torch.empty(B * T, ce, dtype=torch.float32, device=dev)
        ops.call("chanattn_fwd", q=q, kvT=kvT, rawchan=rawchan, cx=cx, B=B, T=T, C=C, ce=ce, nh=nwin, nw=nwin, kv_dtype=F32, ldk=Cp,
                 scale=ce ** -0.5, kvbias=blk.chan_kv.bias.detach() if blk.chan_kv.bias is not None else None)
        if not blk.last_block:
            cp = self._lin(cx, blk.chan_proj, tag + ('cpj',), out_dtype=torch.float32)[0]
            pr = XT2.view(B, N, C)[:, :T]
            self._lin(cp, blk.token_trans1, tag + ('tt1',), out=pr, d_rows=(T, N * C, C), resid=pr, M=B * T)
        # MLP over prompts + pixels (:413, :409)
        xn2, _, _ = ops.layernorm(XT2, blk.norm2.weight.detach(), blk.norm2.bias.detach(), blk.norm2.eps, prec)
        hmid = self._lin(xn2, blk.mlp.fc1, tag + ('fc1',), act=ACT_GELU)[0]
        self._lin(hmid, blk.mlp.fc2, tag + ('fc2',), out=XT2, resid=XT2)
        return XT2, rawlog, rawchan

    def _merge(self, ds, il, XT, rawlog, rawchan, B, T, res, nH):
        prec = self.prec
        dev = XT.device
        H, W = res
        C = ds.dim
        N, N2 = T + H * W, T + (H // 2) * (W // 2)
        tag = ('swm', il)
        cat = torch.empty(B * (N2 - T), 4 * C, dtype=torch.float32, device=dev)
        for k, idx in enumerate(merge_tables(res, T, dev)):
            _gather(XT, cat[:, k * C:], idx, N2 - T, C, C, 4 * C, B, N * C, (N2 - T) * 4 * C)
        catn, _, _ = ops.layernorm(cat, ds.norm.weight.detach(), ds.norm.bias.detach(), ds.norm.eps, prec)
        XT2 = torch.empty(B * N2, 2 * C, dtype=torch.float32, device=dev)
        self._lin(catn, ds.reduction, tag + ('red',), out=XT2.view(B, N2, 2 * C)[:, T:], d_rows=(N2 - T, N2 * 2 * C, 2 * C), M=B * (N2 - T))
        self._lin(XT, ds.task_prompts_up, tag + ('tpu',), a_rows=(T, N * C, C), M=B * T, out=XT2.view(B, N2, 2 * C)[:, :T],
                  d_rows=(T, N2 * 2 * C, 2 * C))
        raw2 = torch.zeros(B, nH, T, N2, dtype=torch.float32, device=dev)
        ops.call("conv3s2_nchw", x=rawlog, w=ds.spa_attn_ds.weight.detach().contiguous(), bias=ds.spa_attn_ds.bias.detach(), y=raw2,
                 B=B, Ci=nH * T, Co=nH * T, H=H, W=W, x_bs=nH * T * N, x_cs=N, x_off=T, y_bs=nH * T * N2, y_cs=N2, y_off=T)
        nwin2 = rawchan.shape[2]
        rc2 = self._lin(rawchan.view(B * T * nwin2, C), ds.process_chan_attn, tag + ('pca',), out_dtype=torch.float32)[0]
        return XT2, raw2, rc2.view(B, T, nwin2, -1)[..., :2 * C].contiguous()

    def _task_features(self, xsrc, xview, rawlog, rawchan, il, B, res, C, hg):
        """cal_task_feature (taskprompter_swin.py:715-777) for all tasks -> [T, B*2h*2w, pitch(F)] activation dtype."""
        p, prec = self.p, self.prec
        names = self.all_tasks
        T = len(names)
        h, w = res
        hw, N = h * w, T + h * w
        tar, F = p.level_embed_dim, p.final_embed_dim
        tarp = ops.pitch(tar)
        nwin = int(math.isqrt(p.chan_nheads))
        sp = self._decoder_split(C)
        mod = ops.modulate(xview, C, N * C, rawlog, rawchan, B, T, N, C, (h, w), (nwin, nwin), prec, hg=hg, split=sp)
        dec_w, dec_b = [], []
        for t in names:
            dec_w += [self.fea_decode_spa[il][t][0].weight, self.fea_decode_chan[il][t][0].weight]
            dec_b += [self.fea_decode_spa[il][t][0].bias, self.fea_decode_chan[il][t][0].bias]
        Wdec = ops.pack_linear_split(dec_w, ('swdec', il)) if sp else ops.pack_linear(dec_w, prec, ('swdec', il))
        bdec = ops.stack_vec(dec_b, ('swdecb', il))
        cat = ops.Split.empty((T, B * hw, 2 * tarp), xsrc.device) if sp else torch.empty(T, B * hw, 2 * tarp, dtype=prec.adt, device=xsrc.device)
        ops.linear(mod, Wdec, tar, prec, bias=bdec, out=cat, batch_inner=2, d_z=(B * hw * 2 * tarp, tarp), ldd=2 * tarp, n_store=tarp)
        del mod
        f0 = [self.fea_fuse[il][t][0].weight for t in names]

        def build_f0():
            with torch.no_grad():
                buf = torch.zeros(T, F, 2 * tarp, dtype=torch.float32, device=f0[0].device)
                for i, wt in enumerate(f0):
                    w2 = wt.detach().reshape(F, 2 * tar)
                    buf[i, :, :tar] = w2[:, :tar]
                    buf[i, :, tarp:tarp + tar] = w2[:, tar:]
                return buf.to(prec.adt)
        if sp:
            W0 = ops.pack_kmap_split(f0, F, 2 * tarp, [(0, 0, tar), (tarp, tar, tar)], ('swf0', il))
        else:
            W0 = ops._cached(('swf0', il, prec.name, tuple(id(q) for q in f0)), f0, build_f0)
        b0 = ops.stack_vec([self.fea_fuse[il][t][0].bias for t in names], ('swf0b', il))
        y0 = ops.linear(cat, W0, F, prec, bias=b0, out_dtype=torch.float32 if sp else None)   # 1x1s before the resize: they commute with it
        del cat
        y0 = ops.bilinear(y0, B, y0.shape[-1], h, w, 2 * h, 2 * w, prec.adt)        # :737 / :765
        ff = [self.fea_fuse[il][t] for t in names]
        Wc = ops.pack_conv3([m[1].weight for m in ff], prec, ('swf1', il))
        bns = [m[2] for m in ff]
        if self.training:
            from . import bn as bn_mod
            y1 = ops.conv3x3(y0, Wc, F, F, B, 2 * h, 2 * w, prec, bias=ops.stack_vec([m[1].bias for m in ff], ('swf1b', il)))
            y1 = bn_mod.train_forward(y1, F, list(bns), ACT_GELU)[0]
        else:
            sc, sh = bn_fold(bns, [m[1].bias for m in ff], ('swf2', il))
            y1 = ops.conv3x3(y0, Wc, F, F, B, 2 * h, 2 * w, prec, bias=sh, colscale=sc, act=ACT_GELU)
        W4 = ops.pack_conv3([m[4].weight for m in ff], prec, ('swf4', il))
        return ops.conv3x3(y1, W4, F, F, B, 2 * h, 2 * w, prec, bias=ops.stack_vec([m[4].bias for m in ff], ('swf4b', il)))


def _split_k(K, target=1152):
    """Largest divisor of K that is a multiple of 8 and <= target (K itself when there is none): the K slice of one workgroup batch."""
    best = K
    for ks in range(8, min(K, target) + 1, 8):
        if K % ks == 0:
            best = ks
    return best if best <= target else K


def taskprompter_create_swin_transformer(variant, pretrained=False, default_cfg=None, **kwargs):
    """taskprompter_swin.py:820-839 (no download: load a checkpoint with load_state_dict)."""
    if pretrained:
        raise RuntimeError('pretrained ImageNet weights need network access; construct with pretrained=False and load a checkpoint')
    kwargs.pop('pretrained_strict', None)
    kwargs.pop('num_classes', None)
    model = TaskPrompterSwin(**kwargs)
    model.default_cfg = dict(default_cfg or {}, variant=variant)
    return model


def taskprompter_swin_base_patch4_window12_384(pretrained=False, **kwargs):
    """TaskPrompter on Swin-B @ 384 (taskprompter_swin.py:841-846)."""
    model_kwargs = dict(patch_size=4, window_size=12, embed_dim=128, depths=(2, 2, 18, 2), num_heads=(4, 8, 16, 32), **kwargs)
    return taskprompter_create_swin_transformer('swin_base_patch4_window12_384', pretrained=pretrained, **model_kwargs)
