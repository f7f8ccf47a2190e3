"""Training path of InvPT (InvPT/models/transformers/vit.py, transformer_decoder.py, invpt.py,
InvPT/models/transformer_net.py): the schedule of invpt.py's no-grad forward rebuilt from
torch.autograd.Functions whose forward AND backward run on the libmtt_hip.so kernels.

Shared with the TaskPrompter training path (autograd_path.py): LayerNormFn, AttnHalfFn (no prompt rows),
MlpHalfFn, BLinearFn, Conv3x3Fn (dilated, bias-free), BnActStackFn, BilinearFn.  New here: the ViT patch embed with a
class token, ConvTranspose2d(3, s2) = GEMM + gather, depthwise stride-2 conv, ceil-mode average pooling and the
materialised (2-head, head dim D/2) cross-task attention: scores / softmax / P.V with hand-written backward GEMMs.

torch ops are used only as plumbing between Functions (permuted copies between task-major and batch-major
token order, fp32 residual adds, and the 2-head 1x1 `fuse_attn` mix of invpt.py:208-229, whose upsampling is the
bilinear kernel); dims must be multiples of 8 (true for every published config: 576/288/144, heads 2).
"""
import torch
from torch.autograd import Function

from . import ops
from ._lib import ACT_NONE, ACT_RELU, F32, OP_K, OP_R, dtype_code
from . import autograd_path
from .autograd_path import (AttnHalfFn, MlpHalfFn, BLinearFn, BilinearFn, Conv3x3Fn, ConvHeadFn, LayerNormFn, TaskHeadsFn, _bn_act,
                            _colsum, _dgrad, _gemm, _wgrad)

pitch = ops.pitch


# =================================================================================================
class VitEmbedFn(Function):
    """patchify + k=s=16 conv as GEMM + pos-embed add, class token in row 0 (vit.py:326-333)."""

    @staticmethod
    def forward(ctx, img, Wpe, bpe, pos, cls, geo, prec):
        B, N, hw = geo
        C = Wpe.shape[0]
        XT = torch.empty(B * N, C, dtype=torch.float32, device=img.device)
        XT.view(B, N, C)[:, :1] = cls + pos[:, :1]
        cols = ops.patchify(img.float(), prec)
        ops.linear(cols, ops.pack_linear([Wpe], prec, 'vpe'), C, prec, bias=bpe[None], out=XT.view(B, N, C)[:, 1:],
                   d_rows=(hw, N * C, C), resid=pos[0, 1:], r_rows=(hw, 0, C), M=B * hw)
        ctx.save_for_backward(cols)
        ctx.geo, ctx.prec, ctx.wshape = geo, prec, Wpe.shape
        return XT

    @staticmethod
    def backward(ctx, dXT):
        (cols,) = ctx.saved_tensors
        B, N, hw = ctx.geo
        C = ctx.wshape[0]
        d3 = dXT.contiguous().view(B, N, C)
        dpatch = d3[:, 1:].reshape(B * hw, C)
        dW = _wgrad(dpatch, cols, C, 768, ctx.prec.bwd).view(ctx.wshape)
        db = _colsum(dpatch, C)
        dpos = d3.sum(0, keepdim=True)
        return None, dW, db, dpos, d3[:, :1].sum(0, keepdim=True), None, None


class ConvT3x3s2Fn(Function):
    """ConvTranspose2d(k=3, s=2, p=1, output_padding=1) (transformer_decoder.py:60): one GEMM producing the 9 tap
    products per input pixel, then a gather (+bias) into the 2x map.  x [B*H*W, C] -> [B*2H*2W, pitch(Co)]."""

    @staticmethod
    def forward(ctx, x, weight, bias, geo, prec):
        B, H, W = geo
        C, Co = weight.shape[0], weight.shape[1]
        Cop = pitch(Co)

        def build():
            with torch.no_grad():
                buf = torch.zeros(9, Cop, C, dtype=torch.float32, device=x.device)
                buf[:, :Co] = weight.detach().permute(2, 3, 1, 0).reshape(9, Co, C)           # [ci, co, ky, kx] -> [tap, co, ci]
                return ops.pack_matrix(buf.reshape(9 * Cop, C), prec)[None]
        wall = ops._cached(('se0', prec.name, id(weight)), [weight], build)
        yall = ops.linear(x, wall, 9 * Cop, prec, ldd=9 * Cop)[0]
        bpad = torch.zeros(Cop, dtype=torch.float32, device=x.device)
        bpad[:Co] = bias
        out = torch.empty(B * 4 * H * W, Cop, dtype=prec.adt, device=x.device)
        ops.call("convt3x3s2_gather", yall=yall, out=out, bias=bpad, B=B, H=H, W=W, Cop=Cop, dtype=dtype_code(yall),
                 out_dtype=dtype_code(out))
        ctx.save_for_backward(x, wall)
        ctx.meta = (geo, prec, C, Co, Cop)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, wall = ctx.saved_tensors
        (B, H, W), prec, C, Co, Cop = ctx.meta
        prec = prec.bwd
        dout = dout.contiguous()
        dyall = torch.empty(B * H * W, 9 * Cop, dtype=dout.dtype, device=dout.device)
        ops.call("convt3x3s2_gather_bwd", yall=None, out=None, bias=None, B=B, H=H, W=W, Cop=Cop, dtype=dtype_code(dyall),
                 out_dtype=dtype_code(dout), xargs=[dout, dyall])
        dWall = _wgrad(dyall, x, 9 * Cop, x.shape[1], prec)                                  # [9*Cop, C]
        dweight = dWall.view(9, Cop, -1)[:, :Co, :C].permute(2, 1, 0).reshape(C, Co, 3, 3)
        dx = _dgrad(dyall, wall[0], B * H * W, x.shape[1], 9 * Cop, prec, x.dtype)
        return dx, dweight, _colsum(dout, Co), None, None


class DwConvS2Fn(Function):
    """Per-task depthwise 3x3 stride-2 conv, bias-free (invpt.py:44-52 conv_proj_q).  x [T, B*H*W, D] -> [T, B*Ho*Wo, D]."""

    @staticmethod
    def forward(ctx, x, geo, *ws):
        B, H, W = geo
        T, _, D = x.shape
        wq = torch.stack([w.detach().reshape(D, 9).t() for w in ws], 0).float().contiguous()       # [T, 9, D]
        Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        y = torch.empty(T, B * Ho * Wo, D, dtype=x.dtype, device=x.device)
        ops.call("dwconv3x3s2", x=x, w=wq, y=y, scale=None, shift=None, Z=T, B=B, H=H, W=W, ld=D, dtype=dtype_code(x))
        ctx.save_for_backward(x, wq)
        ctx.geo = geo
        return y

    @staticmethod
    def backward(ctx, dy):
        x, wq = ctx.saved_tensors
        B, H, W = ctx.geo
        T, _, D = x.shape
        dy = dy.contiguous()
        dx, dw = torch.empty_like(x), torch.empty_like(wq)
        ops.call("dwconv3x3s2_bwd", x=x, w=wq, y=None, scale=None, shift=None, Z=T, B=B, H=H, W=W, ld=D, dtype=dtype_code(x),
                 xargs=[dy, dx, dw])
        return (dx, None) + tuple(dw[t].t().reshape(D, 1, 3, 3) for t in range(T))


class AvgPoolFn(Function):
    """AvgPool2d(k, stride k, ceil_mode=True) on NHWC maps (invpt.py:54-66): x [Z, B*H*W, D] -> [Z, B*Ho*Wo, D]."""

    @staticmethod
    def forward(ctx, x, geo):
        B, H, W, k = geo
        Z, _, D = x.shape
        y = torch.empty(Z, B * -(-H // k) * -(-W // k), D, dtype=x.dtype, device=x.device)
        ops.call("avgpool_ceil", x=x, y=y, B=Z * B, H=H, W=W, k=k, ld=D, dtype=dtype_code(x))
        ctx.meta = (geo, x.shape)
        return y

    @staticmethod
    def backward(ctx, dy):
        (B, H, W, k), xshape = ctx.meta
        dy = dy.contiguous()
        dx = torch.empty(xshape, dtype=dy.dtype, device=dy.device)
        ops.call("avgpool_ceil_bwd", x=None, y=None, B=xshape[0] * B, H=H, W=W, k=k, ld=xshape[2], dtype=dtype_code(dy), xargs=[dy, dx])
        return dx, None


def _heads_kw(B, heads):
    return dict(batch=B * heads, batch_inner=heads)


class ScoresFn(Function):
    """S[b, h] = alpha * q[b, :, h] k[b, :, h]^T on batch-major q [B, Q, D], k [B, K, D] (heads = column blocks of
    width D/heads); fp32 scores [B, heads, Q, pitch(K)] (invpt.py:205)."""

    @staticmethod
    def forward(ctx, q, k, heads, alpha, prec):
        B, Q, D = q.shape
        K = k.shape[1]
        hd, Kp = D // heads, pitch(k.shape[1])
        S = torch.empty(B, heads, Q, Kp, dtype=torch.float32, device=q.device)
        _gemm(q, k, S, Q, K, hd, prec, lda=D, ldb=D, ldd=Kp, a_zo=Q * D, a_zi=hd, b_zo=K * D, b_zi=hd, d_zo=heads * Q * Kp,
              d_zi=Q * Kp, alpha=alpha, n_store=Kp, **_heads_kw(B, heads))
        ctx.save_for_backward(q, k)
        ctx.meta = (heads, alpha, prec)
        return S

    @staticmethod
    def backward(ctx, dS):
        q, k = ctx.saved_tensors
        heads, alpha, prec = ctx.meta
        prec = prec.bwd
        B, Q, D = q.shape
        K = k.shape[1]
        hd, Kp = D // heads, pitch(K)
        dS = dS.contiguous()
        dq, dk = torch.empty_like(q), torch.empty_like(k)
        zs = dict(a_zo=heads * Q * Kp, a_zi=Q * Kp, **_heads_kw(B, heads))
        _gemm(dS, k, dq, Q, hd, K, prec, b_op=OP_R, lda=Kp, ldb=D, ldd=D, b_zo=K * D, b_zi=hd, d_zo=Q * D, d_zi=hd, alpha=alpha,
              n_store=hd, **zs)
        _gemm(dS, q, dk, K, hd, Q, prec, a_op=OP_R, b_op=OP_R, lda=Kp, ldb=D, ldd=D, b_zo=Q * D, b_zi=hd, d_zo=K * D, d_zi=hd,
              alpha=alpha, n_store=hd, **zs)
        return dq, dk, None, None, None


class SoftmaxFn(Function):
    """Row softmax over the K valid columns of fp32 scores [B, heads, Q, pitch(K)] -> P (activation dtype)."""

    @staticmethod
    def forward(ctx, S, K, prec):
        P = torch.empty(S.shape, dtype=prec.adt, device=S.device)
        ops.call("softmax_fwd", S=S, P=P, rows=S.numel() // S.shape[-1], cols=K, ld=S.shape[-1], s_dtype=F32, p_dtype=dtype_code(P),
                 scale=1.0)
        ctx.save_for_backward(P)
        ctx.K = K
        return P

    @staticmethod
    def backward(ctx, dP):
        (P,) = ctx.saved_tensors
        dP = dP.contiguous().float()
        dS = torch.zeros_like(dP)
        ops.call("softmax_bwd", P=P, dP=dP, dS=dS, extra=None, rows=P.numel() // P.shape[-1], cols=ctx.K, ld=P.shape[-1], s_dtype=F32,
                 p_dtype=dtype_code(P), scale=1.0, rows_per_mat=P.shape[-2], extra_rows=0, extra_ld=0)
        if P.shape[-1] != ctx.K:
            dS[..., ctx.K:] = 0
        return dS, None, None


class PVFn(Function):
    """o[b, :, h] = P[b, h] v[b, :, h]  -> [B, Q, D] (heads concatenated along channels, invpt.py:234-235)."""

    @staticmethod
    def forward(ctx, P, v, prec):
        B, heads, Q, Kp = P.shape
        K, D = v.shape[1], v.shape[2]
        hd = D // heads
        o = torch.empty(B, Q, D, dtype=prec.adt, device=P.device)
        _gemm(P, v, o, Q, hd, K, prec, b_op=OP_R, lda=Kp, ldb=D, ldd=D, a_zo=heads * Q * Kp, a_zi=Q * Kp, b_zo=K * D, b_zi=hd,
              d_zo=Q * D, d_zi=hd, n_store=hd, **_heads_kw(B, heads))
        ctx.save_for_backward(P, v)
        ctx.prec = prec
        return o

    @staticmethod
    def backward(ctx, do):
        P, v = ctx.saved_tensors
        prec = ctx.prec.bwd
        B, heads, Q, Kp = P.shape
        K, D = v.shape[1], v.shape[2]
        hd = D // heads
        do = do.contiguous()
        dP = torch.empty(B, heads, Q, Kp, dtype=torch.float32, device=P.device)
        _gemm(do, v, dP, Q, K, hd, prec, lda=D, ldb=D, ldd=Kp, a_zo=Q * D, a_zi=hd, b_zo=K * D, b_zi=hd, d_zo=heads * Q * Kp,
              d_zi=Q * Kp, n_store=Kp, **_heads_kw(B, heads))
        dv = torch.empty_like(v)
        _gemm(P, do, dv, K, hd, Q, prec, a_op=OP_R, b_op=OP_R, lda=Kp, ldb=D, ldd=D, a_zo=heads * Q * Kp, a_zi=Q * Kp, b_zo=Q * D,
              b_zi=hd, d_zo=K * D, d_zi=hd, n_store=hd, **_heads_kw(B, heads))
        return dP, dv, None


class FuseAttnFn(Function):
    """Attention message passing (invpt.py:208-229): the previous stage's scores, bilinearly upsampled x2 over each task's query grid,
    and the current scores are mixed over heads by the 1x1 conv `fuse_attn` — one fused kernel forward (mtt_attn_msg), one fused
    kernel backward (mtt_attn_msg_bwd: dcur, the upsampled-space gradient, dW, dbias) + the bilinear-backward gather for dprev."""

    @staticmethod
    def forward(ctx, S, prev, weight, bias, geo):
        B, heads, T, qh, qw, K = geo
        S, prev = S.contiguous(), prev.contiguous()
        w2 = weight.detach().reshape(heads, 2 * heads).contiguous()
        out = torch.empty_like(S)
        ops.call("attn_msg", cur=S, prev=prev, out=out, w=w2, bias=bias.detach(), B=B, heads=heads, T=T, qh=qh, qw=qw, K=K,
                 ldk=S.shape[-1], ldkp=prev.shape[-1])
        ctx.save_for_backward(S, prev, w2)
        ctx.geo, ctx.wshape = geo, weight.shape
        return out

    @staticmethod
    def backward(ctx, dout):
        S, prev, w2 = ctx.saved_tensors
        B, heads, T, qh, qw, K = ctx.geo
        dout = dout.contiguous()
        dcur, dup = torch.empty_like(S), torch.empty_like(S)
        if S.shape[-1] > K:                                            # padded key columns carry no gradient
            dcur[..., K:] = 0
            dup[..., K:] = 0
        dw = torch.empty(heads, 2 * heads, dtype=torch.float32, device=S.device)
        db = torch.empty(heads, dtype=torch.float32, device=S.device)
        ops.call("attn_msg_bwd", cur=S, prev=prev, out=None, w=w2, bias=None, B=B, heads=heads, T=T, qh=qh, qw=qw, K=K,
                 ldk=S.shape[-1], ldkp=prev.shape[-1],
                 xargs=[dout, dcur, dup, dw, db, ops.ws_for("attn_msg_bwd", S.device, B=B, heads=heads, T=T, qh=qh, qw=qw, K=K)])
        Kp = prev.shape[-1]
        dprev = torch.zeros_like(prev)
        ops.call("bilinear_bwd", **{"in": dup}, out=dprev, B=B * heads * T, C=min(Kp, S.shape[-1]), Hin=qh // 2, Win=qw // 2, Hout=qh, Wout=qw,
                 ld_in=Kp, ld_out=S.shape[-1], in_dtype=F32, out_dtype=F32, out_nchw=0, accumulate=1)
        return dcur, dprev, dw.view(ctx.wshape), db, None


# =================================================================================================
class MultiScaleSumFn(Function):
    """acc = sum_i bilinear(y_i -> (th, tw)) over the InvPT stages' task stacks y_i [T, B*gh_i*gw_i, ld] (invpt.py:531-536: every stage's
    normalised features resized to the target resolution and summed) as ONE node: the first resize writes the fp32 sum buffer, the others
    accumulate into it inside the kernel (mtt_resize_desc.accumulate) — no separate [T, B*th*tw, ld] tensor per stage and no add passes
    (2 x 21.6 GB of traffic at the cfg4 batch); a stage that already has the target resolution is added / differentiated as the identity
    it is (its backward hands the incoming gradient through instead of running the adjoint resize)."""

    @staticmethod
    def forward(ctx, geo, *ys):
        B, C, th, tw, sizes = geo
        T, _, ld = ys[0].shape
        acc = torch.empty(T, B * th * tw, ld, dtype=torch.float32, device=ys[0].device)
        for i, (y, (gh, gw)) in enumerate(zip(ys, sizes)):
            assert y.shape[0] == T and y.shape[2] == ld and y.is_contiguous()
            ops.call("bilinear_fwd", **{"in": y}, out=acc, B=T * B, C=ld, Hin=gh, Win=gw, Hout=th, Wout=tw, ld_in=ld, ld_out=ld,
                     in_dtype=dtype_code(y), out_dtype=F32, out_nchw=0, accumulate=1 if i else 0)
        ctx.meta = (geo, [(tuple(y.shape), y.dtype) for y in ys])
        return acc

    @staticmethod
    def backward(ctx, dacc):
        (B, C, th, tw, sizes), metas = ctx.meta
        dacc = dacc.contiguous()
        outs = []
        for (shape, dtype), (gh, gw) in zip(metas, sizes):
            ld = shape[2]
            if (gh, gw) == (th, tw):
                outs.append(dacc if dtype == torch.float32 else ops.cast2d(dacc.view(-1, ld), dacc.numel() // ld, ld, ld, dtype, ldd=ld).view(shape))
                continue
            din = torch.zeros(shape, dtype=torch.float32, device=dacc.device)
            ops.call("bilinear_bwd", **{"in": dacc}, out=din, B=shape[0] * B, C=ld, Hin=gh, Win=gw, Hout=th, Wout=tw, ld_in=ld, ld_out=ld,
                     in_dtype=F32, out_dtype=F32, out_nchw=0, accumulate=1)
            outs.append(din if dtype == torch.float32 else ops.cast2d(din.view(-1, ld), din.numel() // ld, ld, ld, dtype, ldd=ld).view(shape))
        return (None,) + tuple(outs)


def _check8(*dims):
    if any(d != pitch(d) for d in dims):
        raise NotImplementedError("InvPT training path needs channel / head dims that are their own channel pitch (multiples of 8; of 32 from "
                                  f"{ops.PITCH32_FROM} channels on), got {dims}")


def vit_taps(vit, img):
    """Autograd twin of VisionTransformer.forward_taps (vit.py:326-349)."""
    prec = vit.prec
    B = img.shape[0]
    C, nH = vit.embed_dim, vit.num_heads
    hw = vit.patch_embed.num_patches
    N = hw + 1
    XT = VitEmbedFn.apply(img, vit.patch_embed.proj.weight, vit.patch_embed.proj.bias, vit.pos_embed, vit.cls_token, (B, N, hw), prec)
    taps = []
    for i, blk in enumerate(vit.blocks):
        a = blk.attn
        XT2, _, _ = AttnHalfFn.apply(XT, blk.norm1.weight, blk.norm1.bias, blk.norm1.eps, a.qkv.weight, a.qkv.bias, a.proj.weight,
                                     a.proj.bias, None, None, None, None, None, (B, N, nH, 0, 0, 0, 1), prec, ('vblk', i))
        XT = MlpHalfFn.apply(XT2, blk.norm2.weight, blk.norm2.bias, blk.norm2.eps, blk.mlp.fc1.weight, blk.mlp.fc1.bias,
                             blk.mlp.fc2.weight, blk.mlp.fc2.bias, None, (B, N, 0), prec, ('vblk', i))
        if (i + 1) in vit.select_list:
            taps.append(XT.view(B, N, C)[:, 1:].to(prec.adt).reshape(B * hw, C))
    xf = LayerNormFn.apply(XT, vit.norm.weight, vit.norm.bias, vit.norm.eps, prec, None)
    taps.append(xf.view(B, N, C)[:, 1:].reshape(B * hw, C))
    return taps


def _to_batch_major(x, T, B, n):
    """[1, T*B*n, D] task-major rows -> [B, T*n, D]."""
    D = x.shape[-1]
    return x.view(T, B, n, D).permute(1, 0, 2, 3).reshape(B, T * n, D)


def _block(dec, blk, si, Xf, B, T, D, gh, gw, prev_score):
    """InvPTBlock (invpt.py:290-312) on task-major fp32 tokens Xf [T, B*g*g, D]."""
    prec, at = dec.prec, blk.attn
    heads = at.num_heads
    _check8(D, D // heads)
    rows = B * gh * gw
    qh, qw = (gh - 1) // 2 + 1, (gw - 1) // 2 + 1
    kk = 2 ** (si + 1)
    nq, nk = qh * qw, -(-gh // kk) * -(-gw // kk)
    K = T * nk
    tag = ('ipb', si)
    xn = LayerNormFn.apply(Xf.view(T * rows, D), blk.norm1.weight, blk.norm1.bias, blk.norm1.eps, prec, None).view(T, rows, D)
    qmap = DwConvS2Fn.apply(xn, (B, gh, gw), *[m.conv.weight for m in at.conv_proj_q])
    qmap = _bn_act(qmap, [m.bn for m in at.conv_proj_q], D, ACT_NONE, dec.training)
    kvmap = AvgPoolFn.apply(xn, (B, gh, gw, kk))

    def proj(src, lin, n, name):
        y = BLinearFn.apply(src.reshape(1, -1, D), D, 'plain', None, None, prec, tag + (name,), None, lin.weight, lin.bias)
        return _to_batch_major(y, T, B, n)
    q, k, v = proj(qmap, at.proj_q, nq, 'q'), proj(kvmap, at.proj_k, nk, 'k'), proj(kvmap, at.proj_v, nk, 'v')
    S = ScoresFn.apply(q.contiguous(), k.contiguous(), heads, float(D) ** -0.5, prec)             # scale = full dim (invpt.py:92)
    if prev_score is not None:
        # invpt.py:208-229: previous stage's scores upsampled x2 over the query grid, concatenated over heads, 1x1 conv
        Kp = S.shape[-1]
        ph, pw = (gh // 2 - 1) // 2 + 1, (gw // 2 - 1) // 2 + 1                                   # previous stage's query grid
        assert prev_score.shape[2] == T * ph * pw and prev_score.shape[-1] == Kp
        fa = at.fuse_attn
        S = FuseAttnFn.apply(S, prev_score, fa.weight, fa.bias, (B, heads, T, qh, qw, K))
    P = SoftmaxFn.apply(S, K, prec)
    o = PVFn.apply(P, v.contiguous(), prec)                                                     # [B, T*nq, D]
    o_tm = o.view(B, T, nq, D).permute(1, 0, 2, 3).reshape(1, T * B * nq, D)
    om = BLinearFn.apply(o_tm, D, 'plain', None, None, prec, tag + ('po',), None, at.proj.weight, at.proj.bias).view(T, B * nq, D)
    X2 = Xf + BilinearFn.apply(om, (B, D, qh, qw, gh, gw), torch.float32, False)
    X3 = MlpHalfFn.apply(X2.view(T * rows, D), blk.norm2.weight, blk.norm2.bias, blk.norm2.eps, blk.mlp.fc1.weight, blk.mlp.fc1.bias,
                         blk.mlp.fc2.weight, blk.mlp.fc2.bias, None, (1, T * rows, 0), prec, tag)
    return X3.view(T, rows, D), S


def decoder_forward(dec, taps, B, heads=None):
    """Autograd twin of TransformerDecoder.forward_nhwc -> (features [T, B*8mh*8mw, E], {task: inter_pred [1, B*mh*mw, pitch(n)] fp32});
    heads (the MLPHeads, in task order): -> (their predictions [1, B*8mh*8mw, pitch(n)] fp32 per task, inter_preds) instead of the features."""
    p, prec = dec.p, dec.prec
    names = p.TASKS.NAMES
    T = len(names)
    h, w = p.spatial_dim[-1]
    mh, mw = p.mtt_resolution
    C = p.backbone_channels[-1]
    Ed = dec.embed_dim
    E = Ed + p.PRED_OUT_NUM_CONSTANT
    dims = [E, E // 2, E // 4]
    _check8(C, Ed, *dims)
    training = dec.training
    se0, se1 = dec.scale_embed[0], dec.scale_embed[1]
    back0 = ConvT3x3s2Fn.apply(taps[0], se0.weight, se0.bias, (B, h, w), prec)                   # [B*4hw, dims[2]]
    back1 = Conv3x3Fn.apply(taps[1][None], (B, h, w, dims[1], C), prec, 'se1', se1.weight, se1.bias)[0]

    x = BilinearFn.apply(taps[3][None], (B, C, h, w, mh, mw), prec.adt, False)
    pd = [dec.preliminary_decoder[t] for t in names]
    y = Conv3x3Fn.apply(x.repeat(T, 1, 1), (B, mh, mw, C, C), prec, 'pd0', *[m[0].conv.weight for m in pd], *([None] * T))
    y = _bn_act(y, [m[0].bn1 for m in pd], C, ACT_RELU, training)
    y = Conv3x3Fn.apply(y, (B, mh, mw, Ed, C), prec, 'pd1', *[m[1].conv.weight for m in pd], *([None] * T))
    y = _bn_act(y, [m[1].bn1 for m in pd], Ed, ACT_RELU, training)
    inter, xs = {}, []
    y = y.unbind(0)                                                                        # backward = one stack (no per-slice zero fill + add)
    for i, t in enumerate(names):
        n_out = p.TASKS.NUM_OUTPUT[t]
        ih = dec.intermediate_head[t]
        inter[t] = BLinearFn.apply(y[i][None], n_out, 'plain', None, torch.float32, prec, ('ih', t), None, ih.weight, ih.bias)
        mp = dec.invpt.mix_proj[t][0]                                                      # 1x1 on cat([feature, inter_pred])
        part = BLinearFn.apply(y[i][None], E, 'plain', (Ed, [(0, 0, Ed)]), torch.float32, prec, ('mixa', t), None, mp.weight, mp.bias)
        zero_b = ops._cached(('mixzb', t, id(mp.bias)), [], lambda: torch.zeros(E, dtype=torch.float32, device=mp.bias.device))
        second = BLinearFn.apply(inter[t].to(prec.adt), E, 'plain', (pitch(n_out), [(0, Ed, n_out)]), torch.float32, prec, ('mixb', t), None,
                                 mp.weight, zero_b)
        xs.append(part[0] + second[0])
    Xf = torch.stack(xs, 0)                                                                # fp32 [T, rows0, E]

    th, tw = mh * 8, mw * 8
    scales, scale_sizes = [], []
    prev_score = None
    gh, gw = mh, mw
    for i in range(3):
        D = dims[i]
        stage = dec.invpt.invpt_stages[i]
        if i > 0:
            ue = [m.proj for m in stage.patch_embed]
            Din = dims[i - 1]
            up = BilinearFn.apply(Xf.to(prec.adt), (B, Din, gh, gw, 2 * gh, 2 * gw), prec.adt, False)
            gh, gw = 2 * gh, 2 * gw
            yy = Conv3x3Fn.apply(up, (B, gh, gw, D, Din, 2), prec, ('ue1', i), *[m[1].weight for m in ue], *([None] * T))
            yy = _bn_act(yy, [m[2] for m in ue], D, ACT_RELU, training)
            yy = Conv3x3Fn.apply(yy, (B, gh, gw, D, D, 2), prec, ('ue2', i), *[m[4].weight for m in ue], *([None] * T))
            yy = _bn_act(yy, [m[5] for m in ue], D, ACT_RELU, training)
            skip = back1 if i == 1 else back0                                              # invpt.py:406-411
            Xf = yy.float() + skip.float()[None]
        rows = B * gh * gw
        Xf, prev_score = _block(dec, stage.blocks[0], i, Xf, B, T, D, gh, gw, prev_score)
        # LayerNorm over all tasks' channels (invpt.py:526-530): rows -> [rows, T*D]
        nm = dec.invpt.norm_mts[i]
        cat = Xf.permute(1, 0, 2).reshape(rows, T * D)
        yn = LayerNormFn.apply(cat, nm.weight, nm.bias, nm.eps, prec, None).view(rows, T, D).permute(1, 0, 2).contiguous()
        if i > 0:
            rc = dec.invpt.redu_chan[i]
            yn = BLinearFn.apply(yn, E, 'plain', None, None, prec, ('rc', i), None, *[m.weight for m in rc], *[m.bias for m in rc])
        scales.append(yn)
        scale_sizes.append((gh, gw))
    acc = MultiScaleSumFn.apply((B, E, th, tw, tuple(scale_sizes)), *scales)
    mps = [dec.invpt.mt_proj[t] for t in names]
    if heads is not None and autograd_path.FUSE_HEAD_NODE:
        # mt_proj (Conv3x3 + BatchNorm + ReLU, invpt.py:538-541) and the MLPHeads' 1x1 predictions as ONE node: the 128 x 128 gradient maps
        # of the six tasks stay in the backward's dtype (autograd_path.ConvHeadFn)
        bns = [m[1] for m in mps]
        preds = ConvHeadFn.apply(acc.to(prec.adt), ('conv', (B, th, tw, E, E), 'mtp', E, ACT_RELU, 'iph'), prec, training, bns,
                                 *[m[0].weight for m in mps], *[m[0].bias for m in mps], *[bn.weight for bn in bns], *[bn.bias for bn in bns],
                                 *[hd.linear_pred.weight for hd in heads], *[hd.linear_pred.bias for hd in heads])
        return preds, inter
    f = Conv3x3Fn.apply(acc.to(prec.adt), (B, th, tw, E, E), prec, 'mtp', *[m[0].weight for m in mps], *[m[0].bias for m in mps])
    f = _bn_act(f, [m[1] for m in mps], E, ACT_RELU, training)
    if heads is not None:
        return TaskHeadsFn.apply(f, prec, 'iph', *[hd.linear_pred.weight for hd in heads], *[hd.linear_pred.bias for hd in heads]), inter
    return f, inter


def net_forward(net, x):
    """Autograd twin of TransformerNet.forward (transformer_net.py:25-38)."""
    img_size = tuple(x.shape[-2:])
    B = x.shape[0]
    dec = net.multi_task_decoder
    prec = dec.prec
    taps = vit_taps(net.backbone, x)
    hds = [net.heads[t] for t in net.tasks]
    preds, inter = decoder_forward(dec, taps, B, heads=hds)
    mh, mw = net.p.mtt_resolution
    th, tw = 8 * mh, 8 * mw
    out = {}
    for t, hd, pred in zip(net.tasks, hds, preds):
        out[t] = BilinearFn.apply(pred, (B, hd.linear_pred.weight.shape[0], th, tw, img_size[0], img_size[1]), torch.float32, True)
    out['inter_preds'] = {t: BilinearFn.apply(inter[t], (B, net.p.TASKS.NUM_OUTPUT[t], mh, mw, img_size[0], img_size[1]),
                                              torch.float32, True) for t in net.tasks}
    return out
