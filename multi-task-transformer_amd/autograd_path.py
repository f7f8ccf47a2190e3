"""Training path of TaskPrompter: the same HIP kernels wrapped as torch.autograd.Functions with
hand-written backward passes (dgrad / wgrad are the same MFMA GEMM kernel with transposed operand
views: MTT_OP_R; conv dgrad = the implicit-GEMM conv with mirrored taps; conv wgrad = MTT_OP_CONV_R).

The forward schedule is identical to TaskPrompter._forward_nograd; autograd only carries the fp32
residual stream, the logit side channels and the decoder feature maps between Functions.  Round 1
attention backward materialises P per (batch, head) with the batched GEMM + row-softmax kernels
(no N x N tensor ever leaves the backward), see DESIGN.md.
"""
import math

import torch
import torch.distributed as dist
import torch.nn as nn
from torch.autograd import Function

from . import bn as bn_mod
from . import _lib, ops
from ._lib import ACT_GELU, ACT_GELU_BWD, ACT_GELU_DAUX, ACT_MUL_AUX, ACT_NONE, F32, OP_CONV_R, OP_K, OP_R, dtype_code  # noqa: F401

pitch = ops.pitch
SPLITK_MIN_ROWS = 4096      # reduction length from which few-tile weight gradients are split over the batch dimension
WGRAD_MAX_SLICES, WGRAD_TARGET_WGS = 192, 512        # _wgrad's split-K: up to two workgroups per CU (round 4: 64, 256)


def _gemm(A, B, D, M, N, K, prec, **kw):
    args = dict(A=A, B=B, D=D, M=M, N=N, K=K, a_op=OP_K, b_op=OP_K, a_dtype=dtype_code(A), b_dtype=dtype_code(B),
                d_dtype=dtype_code(D), prec=prec.code, batch=1, batch_inner=1, alpha=1.0)
    args.update(kw)
    ops.call("gemm", **args)
    return D


def _colsum(x2d, cols):
    return ops.colsum(x2d, cols)


def _scaled(g, rowscale, mb, n_prompt, prec):
    """g * per-row DropPath scale, cast to the activation dtype (identity when no DropPath)."""
    if rowscale is None and (g.dtype == prec.adt or not FAST_BWD):
        return g
    out = torch.empty(g.shape, dtype=prec.adt, device=g.device)
    ops.call("rowscale_cast", args=[g, out, g.shape[0], g.shape[1], g.stride(0), out.stride(0), dtype_code(g), dtype_code(out),
                                    rowscale, mb, n_prompt])
    return out


def _scaled_colsum(g, rowscale, mb, n_prompt, prec):
    """-> (_scaled(g, ...), column sums of the result as stored): g is the gradient of a Linear's output (proj / fc2 write the residual
    stream), so the cast that prepares it for the bf16 GEMMs yields that layer's bias gradient in the same pass (mtt_rowscale_cast_colsum).
    Falls back to a separate column sum when no cast is needed."""
    cols = g.shape[1]
    if (rowscale is None and (g.dtype == prec.adt or not FAST_BWD)) or cols % 8 or g.stride(0) % 8:
        out = _scaled(g, rowscale, mb, n_prompt, prec)
        return out, _colsum(out, cols)
    out = torch.empty(g.shape, dtype=prec.adt, device=g.device)
    cs = torch.empty(cols, dtype=torch.float32, device=g.device)
    ws = ops.workspace(_lib.load().mtt_rowscale_cast_colsum_ws_floats(g.shape[0], cols), g.device)
    ops.call("rowscale_cast_colsum", args=[g, out, g.shape[0], cols, g.stride(0), out.stride(0), dtype_code(g), dtype_code(out),
                                           rowscale, mb, n_prompt, cs, ws])
    return out, cs


def _to_bwd(t, prec):
    """A bf16 copy of an fp32-STORED tensor wherever the backward's arithmetic is bf16 (`prec` = the backward's Prec): the backward GEMMs then
    take the token-major / LDS-DMA kernels instead of the register-staged kernel's fp32 modes (which round the operand to bf16 while
    staging: the same products at a third of the rate).  Concerns x3f's fp32-stored decoder tensors and any fp32 dy reaching a bf16-mode
    layer (e.g. the fp32 head outputs); no-op for tensors that already are bf16 and in x3 (whose backward stays fp32-class).  NOTE the bias
    gradients of these layers are column sums of the ROUNDED dy (bf16-class like the rest of that backward; tolerance 5e-3 in the tests)."""
    if prec.name != "bf16" or t.dtype != torch.float32 or not FAST_BWD:
        return t
    t2 = t.reshape(-1, t.shape[-1])
    return ops.cast_rows(t2, torch.bfloat16).view(t.shape)


def _wgrad(dy, x, N, Kp, prec, rows=None, lda=None, ldb=None):
    """dW[N, Kp] = dy[rows, :N]^T @ x[rows, :Kp]  (both operands row-contiguous views).

    When the output has only a few 128x128 tiles but the reduction (tokens / pixels) is long, the reduction is split
    over the GEMM's batch dimension into fp32 slabs that are summed afterwards (split-K: fills the 256 CUs instead of
    running thousands of K steps in a handful of workgroups)."""
    rows = rows if rows is not None else dy.shape[0]
    lda, ldb = lda or dy.stride(0), ldb or x.stride(0)
    tiles = -(-N // 128) * -(-Kp // 128)
    if tiles < 96 and rows >= SPLITK_MIN_ROWS:
        # two workgroups per CU (the kernel's LDS allows it): a handful of output tiles over a million pixel rows (the head predictions'
        # weight gradients: 3 tiles) is pure streaming of x — 64 slices left three quarters of the CUs without a workgroup
        Z = max(2, min(WGRAD_MAX_SLICES, WGRAD_TARGET_WGS // tiles, max(2, rows // 512)))
        c = rows // Z
        rem = rows - c * Z
        slabs = torch.empty(Z + (1 if rem else 0), N, Kp, dtype=torch.float32, device=dy.device)
        _gemm(dy, x, slabs, N, Kp, c, prec, a_op=OP_R, b_op=OP_R, lda=lda, ldb=ldb, ldd=Kp, batch=Z, a_zo=c * lda, b_zo=c * ldb,
              d_zo=N * Kp)
        if rem:
            _gemm(dy[c * Z:], x[c * Z:], slabs[Z], N, Kp, rem, prec, a_op=OP_R, b_op=OP_R, lda=lda, ldb=ldb, ldd=Kp)
        return slabs.sum(0)
    dW = torch.empty(N, Kp, dtype=torch.float32, device=dy.device)
    return _gemm(dy, x, dW, N, Kp, rows, prec, a_op=OP_R, b_op=OP_R, lda=lda, ldb=ldb, ldd=Kp)


SPLIT_TARGET_WGS, SPLIT_ROW_UNIT = 1024, 512        # (tests lower SPLIT_ROW_UNIT to exercise the sliced paths on miniature maps)


def _n_splits(tiles, units, max_splits=16):
    """Number of reduction slices for a weight-gradient GEMM with `tiles` output tiles (all batches) whose reduction has `units`
    indivisible units (row blocks, or images for the 3x3 conv): enough slices to give every CU a few workgroups."""
    if tiles >= SPLIT_TARGET_WGS * 3 // 4 or units < 2:
        return 1
    return max(1, min(max_splits, units, -(-SPLIT_TARGET_WGS // tiles)))


def _wgrad_batched(dy, x, N, Kp, M, prec, Z, lda, ldb, a_zo, b_zo, a0=0, b0=0):
    """dW[z] = dy[z]^T x[z] for z < Z (row-contiguous operands, element offsets a0 + z*a_zo / b0 + z*b_zo): one launch over
    (task, reduction slice) with fp32 slabs summed afterwards when the tile count alone cannot fill the chip."""
    A = dy.reshape(-1)[a0:] if a0 else dy
    B = x.reshape(-1)[b0:] if b0 else x
    p256 = (-(-N // 256)) * (-(-Kp // 256))
    if (prec.name == "bf16" and FAST_BWD and dy.dtype == torch.bfloat16 and x.dtype == torch.bfloat16 and M >= SPLITK_MIN_ROWS
            and min(N, Kp) >= FAST_MIN_DIM and 100 * N * Kp >= 45 * p256 * 65536 and lda % 8 == 0 and ldb % 8 == 0):
        # token-major LDS-DMA kernel (the policy of mtt_gemm picks it for these launches): 256 x 256 tiles, slices chosen to fill the
        # 256 CUs in whole rounds.  N = 300 / 350 outputs waste 41-53 % of a tile pair and it still wins: 432 vs 697 us (fea_decode),
        # 313 vs 424 (fea_fuse[0]), 232 vs 286 (fea_fuse[4]) at B = 63 (profiles/r03_dec_wgrad_bench_h.log)
        S = _tn_splits(Z * p256, M)
        c = (M // S) // 64 * 64
        if c >= 512:
            nz = M // c
            rem = M - nz * c
            slabs = torch.empty(Z, nz + (1 if rem else 0), N, Kp, dtype=torch.float32, device=dy.device)
            SS = slabs.shape[1]
            _gemm(A, B, slabs, N, Kp, c, prec, a_op=OP_R, b_op=OP_R, lda=lda, ldb=ldb, ldd=Kp, batch=Z * nz, batch_inner=nz,
                  a_zo=a_zo, a_zi=c * lda, b_zo=b_zo, b_zi=c * ldb, d_zo=SS * N * Kp, d_zi=N * Kp)
            if rem:
                _gemm(dy.reshape(-1)[a0 + nz * c * lda:], x.reshape(-1)[b0 + nz * c * ldb:], slabs[:, nz], N, Kp, rem, prec, a_op=OP_R,
                      b_op=OP_R, lda=lda, ldb=ldb, ldd=Kp, batch=Z, a_zo=a_zo, b_zo=b_zo, d_zo=SS * N * Kp)
            return slabs.sum(1)
    tiles = Z * (-(-N // 128)) * (-(-Kp // 128))
    S = _n_splits(tiles, M // SPLIT_ROW_UNIT)
    if S == 1:
        dW = torch.empty(Z, N, Kp, dtype=torch.float32, device=dy.device)
        _gemm(A, B, dW, N, Kp, M, prec, a_op=OP_R, b_op=OP_R, lda=lda, ldb=ldb, ldd=Kp, batch=Z, a_zo=a_zo, b_zo=b_zo, d_zo=N * Kp)
        return dW
    c = (M // S) // 8 * 8
    rem = M - c * S
    slabs = torch.empty(Z, S + (1 if rem else 0), N, Kp, dtype=torch.float32, device=dy.device)
    SS = slabs.shape[1]
    _gemm(A, B, slabs, N, Kp, c, prec, a_op=OP_R, b_op=OP_R, lda=lda, ldb=ldb, ldd=Kp, batch=Z * S, batch_inner=S,
          a_zo=a_zo, a_zi=c * lda, b_zo=b_zo, b_zi=c * ldb, d_zo=SS * N * Kp, d_zi=N * Kp)
    if rem:
        _gemm(dy.reshape(-1)[a0 + c * S * lda:], x.reshape(-1)[b0 + c * S * ldb:], slabs[:, S], N, Kp, rem, prec, a_op=OP_R, b_op=OP_R,
              lda=lda, ldb=ldb, ldd=Kp, batch=Z, a_zo=a_zo, b_zo=b_zo, d_zo=SS * N * Kp)
    return slabs.sum(1)


def _pad_last(t, width):
    """contiguous copy of `t` with its last dim zero-padded to `width`."""
    out = torch.zeros(t.shape[:-1] + (width,), dtype=t.dtype, device=t.device)
    out[..., :t.shape[-1]] = t
    return out


N_CUS = 256                 # one 256 x 256 weight-gradient tile occupies a whole CU (128 KiB of LDS)


def _tn_splits(tiles, rows, max_splits=32):
    """Reduction slices for a weight-gradient GEMM with `tiles` 256 x 256 output tiles over `rows` tokens.  The launch takes
    ceil(tiles * S / 256) rounds of workgroups that each reduce rows / S tokens (~4.7 TFLOP/s per CU), and every slice costs one fp32
    slab written and read back (~4 TB/s): pick the S with the smallest modelled time (48 tiles: S = 5 -> 240 workgroups in ONE round,
    where ceil(256 / 48) = 6 needs two rounds for 288 workgroups)."""
    best, cost = 2, None
    for S in range(2, max_splits + 1):
        t = -(-tiles * S // N_CUS) / S * rows * 131072 / 4.7e12 + S * tiles * 65536 * 8 / 4e12
        if cost is None or t < cost:
            best, cost = S, t
    return best


def _wgrad_tn(dy, x, N, Kp, prec):
    """dW[N, Kp] = dy^T x straight from the token-major operands (mtt_gemm MTT_OP_R x MTT_OP_R -> gemm_tn_kernel: LDS-DMA of the rows as
    they sit in memory + LDS transpose reads), the token reduction sliced over the batch dimension into fp32 slabs when the 256 x 256
    output tiles alone cannot fill the chip."""
    rows = dy.shape[0]
    lda, ldb = dy.stride(0), x.stride(0)
    tiles = -(-N // 256) * -(-Kp // 256)
    if tiles < 192 and rows >= 4096:
        S = _tn_splits(tiles, rows)
        c = (rows // S) // 64 * 64
        if c >= 512:
            nz = rows // c
            rem = rows - nz * c
            slabs = torch.empty(nz + (1 if rem else 0), N, Kp, dtype=torch.float32, device=dy.device)
            _gemm(dy, x, slabs, N, Kp, c, prec, a_op=OP_R, b_op=OP_R, lda=lda, ldb=ldb, ldd=Kp, batch=nz, a_zo=c * lda, b_zo=c * ldb, d_zo=N * Kp)
            if rem:
                _gemm(dy[nz * c:], x[nz * c:], slabs[nz], N, Kp, rem, prec, a_op=OP_R, b_op=OP_R, lda=lda, ldb=ldb, ldd=Kp)
            return slabs.sum(0)
    dW = torch.empty(N, Kp, dtype=torch.float32, device=dy.device)
    return _gemm(dy, x, dW, N, Kp, rows, prec, a_op=OP_R, b_op=OP_R, lda=lda, ldb=ldb, ldd=Kp)


def _enc_wgrad(dy, x, N, Kp, prec, bias=True):
    """-> (dW [N, Kp], dbias [N]) of y = x W^T + b given dy (bias=False: the caller already has dbias, e.g. from a GEMM epilogue)."""
    db = _colsum(dy, N) if bias else None
    if (prec.name == "bf16" and FAST_BWD and N >= FAST_MIN_DIM and Kp >= FAST_MIN_DIM and dy.shape[0] >= FAST_MIN_ROWS
            and N % 8 == 0 and dy.stride(0) % 8 == 0):
        if dy.dtype == torch.bfloat16 and x.dtype == torch.bfloat16:
            return _wgrad_tn(dy, x, N, Kp, prec), db
    return _wgrad(dy, x, N, Kp, prec), db


def _enc_dgrad(dy, weight, wpack2d, M, N_in, K_out, prec, out_dtype, tag, colsum=False, **epi):
    """dx = dy @ W.  bf16 mode: uses a cached transposed pack W^T [N_in, K_out] so that both operands are
    reduction-contiguous (fast GEMM path); otherwise the transposing B stager.  colsum=True -> (dx, column sums of dx as stored): with the
    GELU' epilogue dx IS the gradient of the previous Linear's output, so its column sums are that layer's bias gradient — taken in the
    GEMM epilogue (mtt_gemm_desc.colsum_out) instead of re-reading the [tokens, hidden] gradient."""
    if colsum:
        # the epilogue column sums exist for bf16 x bf16 operands and for x3; fp32-stored operands under bf16 arithmetic (x3f with
        # FAST_BWD off: gemm_kernel<.., 3>) take a separate column-sum pass over dx instead (mtt_gemm would return MTT_E_UNSUPPORTED)
        if prec.name == "bf16" and (dy.dtype != torch.bfloat16 or wpack2d.dtype != torch.bfloat16):
            dx = _enc_dgrad(dy, weight, wpack2d, M, N_in, K_out, prec, out_dtype, tag, **epi)
            return dx, _colsum(dx, N_in)
        cs = torch.empty(N_in, dtype=torch.float32, device=dy.device)
        epi = dict(epi, colsum_out=cs, colsum_ws=ops.ws_for("gemm_colsum", dy.device, M=M, N=N_in))
        return _enc_dgrad(dy, weight, wpack2d, M, N_in, K_out, prec, out_dtype, tag, **epi), cs
    if prec.name == "bf16" and FAST_BWD and dy.dtype == torch.bfloat16 and K_out % 64 == 0 and N_in >= FAST_MIN_DIM:
        wT = ops.pack_linear_T(weight, torch.bfloat16, tag)
        dx = torch.empty(M, N_in, dtype=out_dtype, device=dy.device)
        return _gemm(dy, wT, dx, M, N_in, K_out, prec, lda=dy.stride(0), ldb=wT.stride(0), ldd=N_in, n_store=N_in, **epi)
    return _dgrad(dy, wpack2d, M, N_in, K_out, prec, out_dtype, **epi)


FAST_BWD = True
GELU_DAUX = True            # MlpHalfFn with a bf16 backward: fc1 stores GELU'(z), the fc2 dgrad epilogue multiplies (tests / A-B runs set it to False)
FAST_MIN_DIM, FAST_MIN_ROWS = 256, 1024      # below these the general (transposing-stager) kernels are used


def _dgrad(dy, wpack2d, M, N_in, K_out, prec, out_dtype, **epi):
    """dx[M, N_in] = dy[M, :K_out] @ W[K_out, N_in]   (W stored [K_out, ldw], read transposed)."""
    dx = torch.empty(M, N_in, dtype=out_dtype, device=dy.device)
    return _gemm(dy, wpack2d, dx, M, N_in, K_out, prec, b_op=OP_R, lda=dy.stride(0), ldb=wpack2d.stride(0), ldd=N_in,
                 n_store=N_in, **epi)


# =================================================================================================
class LayerNormFn(Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, eps, prec, out_dtype):
        y, mean, rstd = ops.layernorm(x, gamma, beta, eps, prec, save_stats=True, out_dtype=out_dtype)
        ctx.save_for_backward(x, gamma, mean, rstd)
        ctx.eps = eps
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, mean, rstd = ctx.saved_tensors
        dy = dy.contiguous()
        dx, dg, db = torch.zeros_like(x), torch.empty_like(gamma), torch.empty_like(gamma)
        ops.call("layernorm_bwd", x=x, dy=dy, gamma=gamma, mean=mean, rstd=rstd, dx=dx, dgamma=dg, dbeta=db,
                 rows=x.shape[0], C=x.shape[1], ldx=x.stride(0), ldy=dy.stride(0), y_dtype=dtype_code(dy), eps=ctx.eps,
                 ws=ops.ln_bwd_ws(x.shape[0], x.shape[1], x.device))
        return dx, dg, db, None, None, None


# =================================================================================================
def attention_bwd(qkv, dao, drawlog, B, N, nH, T, prec):
    """Backward of ops.attention: recompute P per (batch, head) with the batched GEMM + row softmax."""
    C, Np, Z = nH * 64, pitch(N), B * nH
    dev, adt = qkv.device, prec.adt
    scale = 64 ** -0.5
    S = torch.empty(Z, N, Np, dtype=adt, device=dev)
    zS = dict(batch=Z, batch_inner=nH, d_zo=nH * N * Np, d_zi=N * Np)
    zq = dict(a_zo=N * 3 * C, a_zi=64)
    _gemm(qkv, qkv[:, C:], S, N, N, 64, prec, lda=3 * C, ldb=3 * C, ldd=Np, b_zo=N * 3 * C, b_zi=64, n_store=Np, **zq, **zS)
    ops.call("softmax_fwd", S=S, P=S, rows=Z * N, cols=N, ld=Np, s_dtype=dtype_code(S), p_dtype=dtype_code(S), scale=scale)
    dP = torch.empty(Z, N, Np, dtype=adt, device=dev)
    _gemm(dao, qkv[:, 2 * C:], dP, N, N, 64, prec, lda=C, ldb=3 * C, ldd=Np, a_zo=N * C, a_zi=64, b_zo=N * 3 * C, b_zi=64,
          n_store=Np, **zS)
    ops.call("softmax_bwd", P=S, dP=dP, dS=dP, extra=drawlog, rows=Z * N, cols=N, ld=Np, s_dtype=dtype_code(S),
             p_dtype=dtype_code(S), scale=scale, rows_per_mat=N, extra_rows=T if drawlog is not None else 0, extra_ld=N)
    dqkv = torch.empty(B * N, 3 * C, dtype=adt, device=dev)
    zP = dict(batch=Z, batch_inner=nH, a_zo=nH * N * Np, a_zi=N * Np, d_zo=N * 3 * C, d_zi=64)
    # dV = P^T dO ; dQ = dS K ; dK = dS^T Q
    _gemm(S, dao, dqkv[:, 2 * C:], N, 64, N, prec, a_op=OP_R, b_op=OP_R, lda=Np, ldb=C, ldd=3 * C, b_zo=N * C, b_zi=64, **zP)
    _gemm(dP, qkv[:, C:], dqkv, N, 64, N, prec, b_op=OP_R, lda=Np, ldb=3 * C, ldd=3 * C, b_zo=N * 3 * C, b_zi=64, **zP)
    _gemm(dP, qkv, dqkv[:, C:], N, 64, N, prec, a_op=OP_R, b_op=OP_R, lda=Np, ldb=3 * C, ldd=3 * C, b_zo=N * 3 * C, b_zi=64, **zP)
    return dqkv


HEAD_DGRAD_DMA = True   # TaskHeadsFn: the prediction dgrad on the LDS-DMA kernel (tests / A-B runs set it to False)
FLASH_BWD = True        # bf16: mtt_attn_bwd (tests set it to False to exercise the materialised batched-GEMM backward in bf16)


def attention_bwd_flash(qkv, ao, lse, dao, drawlog, B, N, nH, T, prec):
    """bf16 mode: mtt_attn_bwd (recomputes P tile by tile from q, k and the forward's log-sum-exp; no N x N buffer)."""
    dqkv = torch.empty_like(qkv)
    dsum = torch.empty(B, nH, 2, (N + 3) // 4 * 4, dtype=torch.float32, device=qkv.device)      # rowsum(dO*O) and lse*log2e
    ops.call("attn_bwd", qkv=qkv, out=ao, rawlog=None, lse=lse, B=B, N=N, nH=nH, T=T, dtype=dtype_code(qkv), prec=prec.code,
             scale=64 ** -0.5, xargs=[dao, drawlog, dqkv, dsum])
    return dqkv


def _ln_bwd_join(dres, x, dy, gamma, mean, rstd, eps):
    """-> (dres + LayerNorm backward of dy, dgamma, dbeta): the residual-stream gradient joined with the LayerNorm branch's in the
    same pass that computes the latter (one read of dres, one write of the sum).  Out of place: `dres` is the grad_output autograd
    handed to the Function and may be shared with hooks / retain_grad / other consumers, so it is never written."""
    dg, db = torch.empty_like(gamma), torch.empty_like(gamma)
    out = torch.empty_like(dres)
    ops.call("layernorm_bwd", x=x, dy=dy, gamma=gamma, mean=mean, rstd=rstd, dx=out, dx_in=dres, dgamma=dg, dbeta=db,
             rows=x.shape[0], C=x.shape[1], ldx=x.stride(0), ldy=dy.stride(0), y_dtype=dtype_code(dy), eps=eps,
             ws=ops.ln_bwd_ws(x.shape[0], x.shape[1], x.device))
    return out, dg, db


class AttnHalfFn(Function):
    """First half of a ViT block in ONE autograd node: norm1 -> qkv GEMM -> flash attention (+ prompt-row logits) -> proj GEMM +
    residual [-> channel attention on the prompt rows] (taskprompter.py:195-254, :273-276; vit.py:199-202 without prompts).
    Fusing the node removes three [tokens, C] fp32 passes per block that autograd would add: the sum of the two gradients of
    the normalised tokens, the zero-initialised LayerNorm-backward buffer and its sum with the residual gradient — here the
    channel-attention backward accumulates onto the qkv dgrad output, and the LayerNorm backward adds the incoming
    residual gradient in the pass that produces the branch gradient."""

    @staticmethod
    def forward(ctx, XT, g1, b1, eps, Wqkv, bqkv, Wproj, bproj, Wtt, btt, Wtt1, btt1, rowscale, geo, prec, tag):
        B, N, nH, T, h, w, nwin = geo[:7]
        side = geo[7] if len(geo) > 7 else True              # False: no consumer of this block's channel logits (not a tap): skip that pass
        C, hw = nH * 64, h * w
        chan = Wtt is not None
        split = prec.split                       # x3f: x3 products on pre-split planes (LDS-DMA kernel), bf16 backward on the hi planes
        if split:
            xs, mean, rstd = ops.layernorm(XT, g1, b1, eps, prec, save_stats=True, out_dtype="split")
            xp32 = ops.prompt_rows32(xs, B, N, T, C) if chan else None      # the prompt rows (fp32-class) of the channel attention's Linears
            wq = ops.pack_linear_split([Wqkv], tag + ('qkv',))
            wp = ops.pack_linear_split([Wproj], tag + ('proj',))
            qkv = ops.linear(xs, wq, 3 * C, prec, bias=bqkv[None], out_dtype="split")[0]
            ao, rawlog, lse = ops.attention(qkv, B, N, nH, T, prec, want_lse=True)
            xn_b, xn_c = xs.hi, xs                  # backward operand: the bf16 hi plane; the channel attention reads the planes (no fp32 copy)
        else:
            xn, mean, rstd = ops.layernorm(XT, g1, b1, eps, prec, save_stats=True)
            wq = ops.pack_linear([Wqkv], prec, tag + ('qkv',))
            wp = ops.pack_linear([Wproj], prec, tag + ('proj',))
            qkv = ops.linear(xn, wq, 3 * C, prec, bias=bqkv[None])[0]
            flash = FLASH_BWD and prec.name == "bf16" and qkv.dtype == torch.bfloat16
            ao, rawlog, lse = ops.attention(qkv, B, N, nH, T, prec, want_lse=flash)
            xn_b = xn_c = xn
        XT2 = torch.empty_like(XT)
        ops.linear(ao, wp, C, prec, bias=bproj[None], out=XT2, resid=XT, d_rows=(N, N * C, C), rowscale=rowscale, n_prompt=T,
                   M=B * N)
        cq = wt = wt1 = rawchan = None
        if chan:
            wt = ops.pack_linear([Wtt], prec, tag + ('tt',))
            wt1 = ops.pack_linear([Wtt1], prec, tag + ('tt1',))
            if split:
                cq = ops.linear(xp32, wt, hw, prec, bias=btt[None], M=B * T)[0]
            else:
                cq = ops.linear(xn_c, wt, hw, prec, bias=btt[None], a_rows=(T, N * C, C), M=B * T)[0]
            if side:
                rawchan = ops.chan_logits(cq, xn_c, B, T, N, C, (h, w), (nwin, nwin))
            pr = XT2.view(B, N, C)[:, :T]
            ops.linear(cq, wt1, C, prec, bias=btt1[None], out=pr, d_rows=(T, N * C, C), resid=pr, rowscale=rowscale, n_prompt=T,
                       M=B * T)
        ctx.save_for_backward(XT, g1, mean, rstd, xn_b, ops._hi(qkv), ops._hi(ao), ops._hi(wq), ops._hi(wp), rowscale, lse, cq, wt, wt1,
                              xp32 if chan and split else None)
        ctx.geo, ctx.prec, ctx.eps, ctx.chan = geo, prec, eps, chan
        ctx.params = (Wqkv, Wproj, Wtt, Wtt1)
        # the two logit side channels are consumed only at the four taps (cal_task_feature); for the other 20 blocks autograd would hand
        # backward() MATERIALISED ZERO gradients for them — and the channel-attention backward kernel (227 us per block at the benchmark's
        # batch) would run on zeros.  Without materialisation those gradients arrive as None and their branches are skipped.
        ctx.set_materialize_grads(False)
        z = torch.zeros(0, device=XT.device)
        return XT2, (rawlog if rawlog is not None else z), (rawchan if rawchan is not None else z)

    @staticmethod
    def backward(ctx, dXT2, drawlog, drawchan):
        XT, g1, mean, rstd, xn, qkv, ao, wq, wp, rowscale, lse, cq, wt, wt1, xp32 = ctx.saved_tensors
        Wqkv_, Wproj_, Wtt_, Wtt1_ = ctx.params
        B, N, nH, T, h, w, nwin = ctx.geo[:7]
        prec, C, M, hw = ctx.prec.bwd, nH * 64, B * N, h * w
        xn_c = xn              # the rows the channel attention's backward reads: the bf16 hi plane in the x3f mode (its backward IS bf16)
        if dXT2 is None:                                        # (set_materialize_grads(False): only when the block output were unused)
            dXT2 = torch.zeros_like(XT)
        dXT2 = dXT2.contiguous()
        # ---- spatial attention ---------------------------------------------------------------------------------
        g, dbproj = _scaled_colsum(dXT2, rowscale, N, T, prec)
        dWproj, _ = _enc_wgrad(g, ao, C, C, prec, bias=False)
        dao = _enc_dgrad(g, Wproj_, wp[0], M, C, C, prec, prec.adt, 'proj')
        dl = drawlog.contiguous() if (T > 0 and drawlog is not None and drawlog.numel()) else None
        if lse is not None:
            dqkv = attention_bwd_flash(qkv, ao, lse, dao, dl, B, N, nH, T, prec)
        else:
            dqkv = attention_bwd(qkv, dao, dl, B, N, nH, T, prec)
        dWqkv, dbqkv = _enc_wgrad(dqkv, xn, 3 * C, C, prec)
        dxn = _enc_dgrad(dqkv, Wqkv_, wq[0], M, C, 3 * C, prec, torch.float32, 'qkv')
        # ---- channel attention: accumulates onto dxn -------------------------------------------------------------
        dWtt = dbtt = dWtt1 = dbtt1 = None
        if ctx.chan:
            hwp = cq.shape[-1]
            gp = dXT2.view(B, N, C)[:, :T].reshape(B * T, C)                  # tiny copy (B*T rows)
            if rowscale is not None:
                gp = gp * rowscale[:, 0].repeat_interleave(T)[:, None]
            dWtt1 = _wgrad(gp, cq, C, hwp, prec)[:, :hw]
            dbtt1 = _colsum(gp, C)
            # the two input gradients of the prompt-row Linears (M = B*T rows, K = 1024): bf16 operands + transposed packs put them on the
            # LDS-DMA kernel (a third of the register-staged kernel's K-step latency; the fp32 operand was rounded to bf16 while staged anyway)
            fast_tt = prec.name == "bf16" and FAST_BWD and hwp == hw and hw >= FAST_MIN_DIM and C >= FAST_MIN_DIM and C % 64 == 0
            if fast_tt:
                gp16 = ops.cast_rows(gp, torch.bfloat16)
                dcq = _enc_dgrad(gp16, Wtt1_, wt1[0], B * T, hwp, C, prec, torch.float32, 'tt1')
            else:
                dcq = _dgrad(gp, wt1[0], B * T, hwp, C, prec, torch.float32)
            if drawchan is not None and drawchan.numel():
                dq2 = torch.zeros(B * T, hwp, dtype=torch.float32, device=xn.device)
                # one storage type for q and the tokens: in the x3f mode q (fp32, [B*T, hw]: tiny) is rounded to the hi plane's bf16
                cqb = cq if cq.dtype == xn_c.dtype else ops.cast_rows(cq, xn_c.dtype)
                ops.call("chan_logits_bwd", q=cqb, xn=xn_c, rawchan=None, B=B, T=T, N=N, C=C, h=h, w=w, nh=nwin, nw=nwin,
                         dtype=dtype_code(xn_c), ldq=hwp, xargs=[drawchan.contiguous(), dq2, F32, dxn])
                dcq = dcq + dq2                                                # [B*T, hwp] fp32 (tiny)
            xnp = xp32 if xp32 is not None else xn_c.view(B, N, C)[:, :T].reshape(B * T, C)
            dWtt = _wgrad(dcq, xnp, hw, C, prec)[:, :C]
            dbtt = _colsum(dcq, hw)
            dp = dxn.view(B, N, C)[:, :T]
            if fast_tt:
                dcq16 = ops.cast_rows(dcq, torch.bfloat16)
                wT = ops.pack_linear_T(Wtt_, torch.bfloat16, 'tt')                        # [C, hw]
                _gemm(dcq16, wT, dp, B * T, C, hw, prec, lda=hwp, ldb=hw, ldd=C, d_mb=T, d_bs=N * C,
                      resid=dp, r_mb=T, r_bs=N * C, ldr=C, n_store=C)
            else:
                _gemm(dcq, wt[0], dp, B * T, C, hw, prec, b_op=OP_R, lda=hwp, ldb=wt.shape[-1], ldd=C, d_mb=T, d_bs=N * C,
                      resid=dp, r_mb=T, r_bs=N * C, ldr=C, n_store=C)
        # ---- norm1 backward accumulated into the residual gradient ------------------------------------------------
        dXT, dg1, db1 = _ln_bwd_join(dXT2, XT, dxn, g1, mean, rstd, ctx.eps)
        return (dXT, dg1, db1, None, dWqkv, dbqkv, dWproj, dbproj, dWtt, dbtt, dWtt1, dbtt1, None, None, None, None)


class MlpHalfFn(Function):
    """Second half of a ViT block in one autograd node: norm2 -> fc1 + GELU -> fc2 + residual (taskprompter.py:277; vit.py:203)."""

    @staticmethod
    def forward(ctx, XT2, g2, b2n, eps, W1, b1, W2, b2, rowscale, geo, prec, tag):
        B, N, T = geo
        C, Hd = W1.shape[1], W1.shape[0]
        # the split-plane kernel: whole 32-deep K steps (InvPT's 288-channel stage included; else register-staged x3).  MLP_SPLIT_RULE64: the rule of
        # rounds 3-5 (multiples of 64), for A/B runs
        split = prec.split and ((C % 64 == 0 and Hd % 64 == 0) if MLP_SPLIT_RULE64 else (ops.split_gemm_ok(C) and ops.split_gemm_ok(Hd)))
        xn2, mean, rstd = ops.layernorm(XT2, g2, b2n, eps, prec, save_stats=True, out_dtype="split" if split else None)
        w1 = ops.pack_linear_split([W1], tag + ('fc1',)) if split else ops.pack_linear([W1], prec, tag + ('fc1',))
        w2 = ops.pack_linear_split([W2], tag + ('fc2',)) if split else ops.pack_linear([W2], prec, tag + ('fc2',))
        # what the backward's GELU' needs.  fp32-class backward (x3): the pre-activation z in fp32, GELU'(z) evaluated in the fc2 dgrad
        # epilogue.  bf16 backward (bf16, x3f): GELU'(z) itself, taken HERE where z is in registers in fp32 and stored as bf16 (the same
        # bytes as bf16(z)) — the fc2 dgrad epilogue is then one multiply per element instead of an erf + exp evaluation on 64 890 x 4 096
        # elements per block (GELU_DAUX: the derivative with the forward, round 6)
        z = torch.empty(B * N, Hd, dtype=prec.bwd.adt, device=XT2.device)
        daux = GELU_DAUX and z.dtype == torch.bfloat16
        hmid = ops.linear(xn2, w1, Hd, prec, bias=b1[None], act=ACT_GELU_DAUX if daux else ACT_GELU, aux_out=z, out_dtype="split" if split else None)[0]
        XT3 = torch.empty_like(XT2)
        ops.linear(hmid, w2, C, prec, bias=b2[None], out=XT3, resid=XT2, d_rows=(N, N * C, C), rowscale=rowscale, n_prompt=T,
                   M=B * N)
        ctx.save_for_backward(XT2, g2, mean, rstd, ops._hi(xn2), z, ops._hi(hmid), ops._hi(w1), ops._hi(w2), rowscale)
        ctx.geo, ctx.prec, ctx.eps, ctx.daux = geo, prec, eps, daux
        ctx.params = (W1, W2)
        return XT3

    @staticmethod
    def backward(ctx, dXT3):
        XT2, g2, mean, rstd, xn2, z, hmid, w1, w2, rowscale = ctx.saved_tensors
        B, N, T = ctx.geo
        prec, M = ctx.prec.bwd, B * N
        C, Hd = xn2.shape[1], z.shape[1]
        dXT3 = dXT3.contiguous()
        g, db2 = _scaled_colsum(dXT3, rowscale, N, T, prec)
        W1_, W2_ = ctx.params
        dW2, _ = _enc_wgrad(g, hmid, C, Hd, prec, bias=False)
        dz, db1 = _enc_dgrad(g, W2_, w2[0], M, Hd, C, prec, prec.adt, 'fc2', colsum=True, act=ACT_MUL_AUX if ctx.daux else ACT_GELU_BWD, aux_in=z,
                             aux_dtype=dtype_code(z), ldaux=Hd)
        dW1, _ = _enc_wgrad(dz, xn2, Hd, C, prec, bias=False)
        dxn2 = _enc_dgrad(dz, W1_, w1[0], M, C, Hd, prec, prec.adt, 'fc1')
        dXT2, dg2, dbn2 = _ln_bwd_join(dXT3, XT2, dxn2, g2, mean, rstd, ctx.eps)
        return dXT2, dg2, dbn2, None, dW1, db1, dW2, db2, None, None, None, None


class PatchEmbedFn(Function):
    """patchify + k=s=16 conv as GEMM + pos-embed add, prompts copied in front (taskprompter.py:393-397)."""

    @staticmethod
    def forward(ctx, img, Wpe, bpe, pos, prompts, geo, prec):
        B, N, T, hw = geo
        C = Wpe.shape[0]
        XT = torch.empty(B * N, C, dtype=torch.float32, device=img.device)
        XT.view(B, N, C)[:, :T] = prompts
        cols = ops.patchify(img.float(), prec)
        wpe = ops.pack_linear([Wpe], prec, 'pe')
        ops.linear(cols, wpe, C, prec, bias=bpe[None], out=XT.view(B, N, C)[:, T:], d_rows=(hw, N * C, C),
                   resid=pos[0, 1:], r_rows=(hw, 0, C), M=B * hw)
        ctx.save_for_backward(cols)
        ctx.geo, ctx.prec, ctx.wshape = geo, prec, Wpe.shape
        return XT

    @staticmethod
    def backward(ctx, dXT):
        (cols,) = ctx.saved_tensors
        B, N, T, hw = ctx.geo
        C = ctx.wshape[0]
        d3 = dXT.contiguous().view(B, N, C)
        dpatch = d3[:, T:].reshape(B * hw, C)
        dW = _wgrad(dpatch, cols, C, 768, ctx.prec.bwd).view(ctx.wshape)
        db = _colsum(dpatch, C)
        dpos = torch.zeros(1, hw + 1, C, dtype=torch.float32, device=dXT.device)
        dpos[0, 1:] = d3[:, T:].sum(0)
        return None, dW, db, dpos, d3[:, :T].sum(0), None, None


class ModulateFn(Function):
    """cal_task_feature's (1 + logit) modulation for all tasks (taskprompter.py:436-467)."""

    @staticmethod
    def forward(ctx, xsrc, rawlog, rawchan, geo, prec, split=False):
        """split=True (x3f): the modulated copies as hi / lo bf16 planes (returned as two tensors; lo carries no gradient) for the
        split-plane fea_decode GEMM — the hi plane is also the bf16 operand of its weight gradient, so no cast pass in the backward."""
        B, N, T, C, h, w, nwin = geo[:7]
        hg = geo[7] if len(geo) > 7 else 0               # channels per attention head (0 = 64; the Swin stages pass theirs)
        mod = ops.modulate(xsrc.view(B, N, C)[:, T:], C, N * C, rawlog, rawchan, B, T, N, C, (h, w), (nwin, nwin), prec, hg=hg, split=split)
        ctx.save_for_backward(xsrc, rawlog, rawchan)
        ctx.geo = geo
        if split:
            # the lo plane carries no gradient; without set_materialize_grads(False) autograd would hand backward() a materialised ZERO
            # tensor of its shape anyway (1.6 GB per tap at the benchmark's batch: 0.85 ms of fills per step, profiles/r06_torch_ops_g_*.log)
            ctx.mark_non_differentiable(mod.lo)
            ctx.set_materialize_grads(False)
            return mod.hi, mod.lo
        return mod

    @staticmethod
    def backward(ctx, dmod, *_):
        xsrc, rawlog, rawchan = ctx.saved_tensors
        B, N, T, C, h, w, nwin = ctx.geo[:7]
        hg = ctx.geo[7] if len(ctx.geo) > 7 else 0
        dmod = dmod.contiguous()
        dx = torch.empty_like(xsrc)                                           # patch rows are WRITTEN by the kernel; the T prompt rows get no gradient here
        dx.view(B, N, C)[:, :T].zero_()
        dl, dc = torch.zeros_like(rawlog), torch.empty_like(rawchan)          # drawlog: the first T columns stay zero; drawchan is written
        ops.call("modulate_bwd", x=xsrc.view(B, N, C)[:, T:], x_ld=C, x_bs=N * C, rawlog=rawlog, rawchan=rawchan, out=None,
                 B=B, T=T, N=N, C=C, h=h, w=w, nh=nwin, nw=nwin, out_dtype=dtype_code(dmod), hg=hg,
                 xargs=[dmod, dx.view(B, N, C)[:, T:], dl, dc, ops.ws_for("modulate_bwd", dmod.device, B=B, T=T, C=C, h=h, w=w, nh=nwin, nw=nwin)])
        return dx, dl, dc, None, None, None


# =================================================================================================
MLP_SPLIT_RULE64 = False
AUTO_SPLIT = True             # A/B switch of BLinearFn's own split pass (x3f, fp32 inputs)
AUTO_SPLIT_MIN_ROWS = 2048


class BLinearFn(Function):
    """Task-batched linear / 1x1 conv: y[z] = x[z] @ W[z]^T + b[z].  layout 'plain' -> [Z, M, pitch(N)];
    'catpair' -> [Z/2, M, 2*pitch(N)] (z = 2t+s written at column offset s*pitch(N): the reference's
    torch.cat([spa, chan], dim=1), taskprompter.py:471).  kmap = optional (Kp, [(dst0, src0, len), ...])
    column remap used when the input is such a padded concatenation."""

    @staticmethod
    def forward(ctx, x, N, layout, kmap, out_dtype, prec, tag, xlo, *wb):
        """xlo: None, or the lo plane of a split input (x is then its hi plane): the product runs on the split-plane LDS-DMA kernel with
        pre-split weight planes.  out_dtype "split": the output as (hi, lo) planes — two tensors, lo without gradient."""
        Z = len(wb) // 2
        ws, bs = wb[:Z], wb[Z:]
        xin = ops.Split(x, xlo) if xlo is not None else x
        # x3f with an fp32 input nobody split yet (the Swin / InvPT Linears: window tokens, attention outputs, merged patches): one
        # split pass, then the split-plane LDS-DMA kernel instead of the register-staged x3 one (which splits every tile of both operands
        # while staging: 8.6 % of the Swin-B step); the hi plane is saved as the backward's bf16 operand instead of the fp32 rows
        auto = (xlo is None and AUTO_SPLIT and prec.split and x.dtype == torch.float32 and x.is_contiguous() and layout == 'plain'
                and kmap is None and out_dtype != "split" and x.shape[-2] >= AUTO_SPLIT_MIN_ROWS
                and ops.split_gemm_ok(ops._w2d(ws[0])[1]) and x.shape[-1] == ops._w2d(ws[0])[1])
        if auto:
            sp = ops.split_cast(x.view(-1, x.shape[-1]))
            xin = ops.Split(sp.hi.view(x.shape), sp.lo.view(x.shape))
        if xlo is not None or auto:
            wpack = ops.pack_linear_split(list(ws), tag) if kmap is None else ops.pack_kmap_split(list(ws), N, kmap[0], kmap[1], tag)
        elif kmap is None:
            wpack = ops.pack_linear(list(ws), prec, tag)
        else:
            wpack = ops.pack_kmap(list(ws), N, kmap[0], kmap[1], prec, tag)
        bias = ops.stack_vec(list(bs), (tag, 'b'))
        M, Np = x.shape[-2], pitch(N)
        if layout == 'catpair':
            shape = (Z // 2, M, 2 * Np)
            out = ops.Split.empty(shape, x.device) if out_dtype == "split" else torch.empty(shape, dtype=out_dtype or prec.adt, device=x.device)
            ops.linear(xin, wpack, N, prec, bias=bias, out=out, batch_inner=2, d_z=(M * 2 * Np, Np), ldd=2 * Np, n_store=Np)
        else:
            out = ops.linear(xin, wpack, N, prec, bias=bias, out_dtype=out_dtype)
        ctx.save_for_backward(xin.hi if auto else x, ops._hi(wpack))   # split: the hi planes ARE the bf16 operands of the (bf16) backward
        ctx.meta = (Z, N, layout, kmap, prec, [tuple(w.shape) for w in ws], x.dtype)
        if isinstance(out, ops.Split):
            ctx.mark_non_differentiable(out.lo)
            ctx.set_materialize_grads(False)             # no zero tensor for the lo plane's (non-existent) gradient
            return out.hi, out.lo
        return out

    @staticmethod
    def backward(ctx, dy, *_):
        x, wpack = ctx.saved_tensors
        Z, N, layout, kmap, prec, wshapes, xdt = ctx.meta
        prec = prec.bwd
        x, dy = _to_bwd(x, prec), _to_bwd(dy.contiguous(), prec)
        M, Np, Kp = x.shape[-2], pitch(N), wpack.shape[-1]
        if layout == 'catpair':
            az = dict(batch=Z, batch_inner=2, a_zo=M * 2 * Np, a_zi=Np)
            lda = 2 * Np
        else:
            az = dict(batch=Z, batch_inner=1, a_zo=M * Np)
            lda = Np
        xz = x.stride(0) if (x.dim() == 3 and x.shape[0] > 1) else 0
        bi = az['batch_inner']
        dx = torch.empty(Z, M, Kp, dtype=getattr(ctx, "grad_dtype", None) or xdt, device=x.device)       # (a stage of FuseTailFn: the backward's dtype)
        if prec.name == "bf16" and FAST_BWD and dy.dtype == torch.bfloat16 and M >= FAST_MIN_ROWS and Kp >= 128:
            # dgrad on the LDS-DMA kernels: reduction-contiguous transposed pack W^T.  The reduction runs over pitch(N) columns of dy:
            # the padding columns [N, pitch(N)) meet zero rows of W^T, but 0 * NaN is NaN, so they must hold FINITE values.  Invariant of
            # this file: every producer of a task-stack gradient writes its padding channels as zeros (conv / linear dgrads through
            # n_store = pitch, bn_bwd_apply, ctr_mix, upconv4_gather, cast2d with zero_pad) — never torch.empty garbage.
            wT = _pad_last(wpack.transpose(1, 2), Np)                  # [Z, Kp, pitch(N)]: a few MB, once per backward of this node
            if wT.dtype != torch.bfloat16:                             # x3f: the forward's pack is fp32
                wT = wT.to(torch.bfloat16)
            _gemm(dy, wT, dx, M, Kp, Np, prec, lda=lda, ldb=Np, ldd=Kp, b_zo=wT.stride(0) * bi, b_zi=wT.stride(0) if bi > 1 else 0,
                  d_zo=M * Kp * bi, d_zi=M * Kp if bi > 1 else 0, n_store=Kp, **az)
        else:
            _gemm(dy, wpack, dx, M, Kp, N, prec, b_op=OP_R, lda=lda, ldb=Kp, ldd=Kp, b_zo=wpack.stride(0) * bi,
                  b_zi=wpack.stride(0) if bi > 1 else 0, d_zo=M * Kp * bi, d_zi=M * Kp if bi > 1 else 0, n_store=Kp, **az)
        if Z == 1 and layout == 'plain':
            dW = _wgrad(dy.view(M, lda), x.reshape(M, x.shape[-1]), N, Kp, prec)[None]
        elif layout == 'catpair':
            # z = 2t + s: dy[t][:, s*Np:], x[z]; one (task, slice)-batched launch per s
            halves = [_wgrad_batched(dy, x, N, Kp, M, prec, Z // 2, lda, x.shape[-1], M * lda, 2 * xz, a0=sft * Np, b0=sft * xz) for sft in (0, 1)]
            dW = torch.stack(halves, 1).reshape(Z, N, Kp)
        else:
            dW = _wgrad_batched(dy, x, N, Kp, M, prec, Z, lda, x.shape[-1], M * lda, xz)
        dys = dy.view(-1, M, lda)
        # bias gradients of all Z layers in one launch pair (catpair: one per half, the half's columns start s * Np into every row)
        if layout == 'catpair':
            h0, h1 = (ops.colsum_batched(dys, N, Z // 2, M * lda, col_off=sft * Np) for sft in (0, 1))
            dball = torch.stack([h0, h1], 1).reshape(Z, N)
        else:
            dball = ops.colsum_batched(dys, N, Z, M * lda)
        dbs = list(dball.unbind(0))
        K = math.prod(wshapes[0][1:])
        cols = [(0, 0, K)] if kmap is None else kmap[1]
        dW = dW.contiguous()
        dws = ops.unpack_grads(dW, ('blinear', N, Kp, tuple(cols)), wshapes,
                               lambda src, flat, offs: [ops.segment(src, z * N * Kp + d0, flat, offs[z] + s0, (1, N, ln), (0, Kp, 1), (0, K, 1))
                                                        for z in range(Z) for (d0, s0, ln) in cols],
                               partial=sum(c[2] for c in cols) != K)      # a column range of a shared weight (InvPT mix projections)
        if x.dim() == 3 and x.shape[0] == 1 and Z > 1:
            dx = dx.sum(0, keepdim=True)
        elif x.dim() == 2:
            dx = dx.sum(0)
        return (dx, None, None, None, None, None, None, None) + tuple(dws) + tuple(dbs)


class Conv3x3Fn(Function):
    """Task-batched 3x3 conv (+bias) as implicit GEMM; dgrad = same kernel with mirrored taps on the
    transposed pack; wgrad = MTT_OP_CONV_R (taskprompter.py:362 fea_fuse[1], :692/:706 head convs).
    geo = (B, H, W, Co, Ci[, dilation]); the Z biases may all be None (bias-free convs of the InvPT decoder)."""

    @staticmethod
    def forward(ctx, x, geo, prec, tag, *wb):
        """geo may carry a 7th element: the lo plane of a split input (x is then its hi plane; x3f) — the conv runs on the split-plane
        implicit-GEMM kernel with pre-split weights, and the hi plane is the bf16 operand of the backward."""
        B, H, W, Co, Ci = geo[:5]
        dil = geo[5] if len(geo) > 5 else 1
        xlo = geo[6] if len(geo) > 6 else None
        Z = len(wb) // 2
        ws, bs = wb[:Z], wb[Z:]
        has_bias = bs[0] is not None
        bias = ops.stack_vec(list(bs), (tag, 'b')) if has_bias else None
        xdt = x.dtype
        if xlo is not None:
            y = ops.conv3x3(ops.Split(x, xlo), ops.pack_conv3_split(list(ws), tag), Co, Ci, B, H, W, prec, dil=dil, bias=bias, out_dtype=torch.float32)
        elif prec.split and x.dtype == torch.float32 and ops.split_conv_ok(Ci, Co):
            # x3f on an fp32 input (head convs, Swin, InvPT): split it HERE, so that the hi plane — the bf16 operand the backward needs
            # anyway — is what gets saved instead of the fp32 activation plus a second cast pass in the backward
            Zx, rows, Cp = x.shape
            sp = ops.split_cast(x.contiguous().view(Zx * rows, Cp))
            xs = ops.Split(sp.hi.view(Zx, rows, Cp), sp.lo.view(Zx, rows, Cp))
            y = ops.conv3x3(xs, ops.pack_conv3_split(list(ws), tag), Co, Ci, B, H, W, prec, dil=dil, bias=bias)
            x = xs.hi
        else:
            y = ops.conv3x3(x, ops.pack_conv3(list(ws), prec, tag), Co, Ci, B, H, W, prec, dil=dil, bias=bias)
        ctx.save_for_backward(x, *ws)
        ctx.meta = ((B, H, W, Co, Ci, dil), prec, tag, Z, has_bias, xdt)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, ws = ctx.saved_tensors[0], ctx.saved_tensors[1:]
        (B, H, W, Co, Ci, dil), prec, tag, Z, has_bias, xdt = ctx.meta
        prec = prec.bwd
        x, dy = _to_bwd(x, prec), _to_bwd(dy.contiguous(), prec)
        rows, Cip, Cop = x.shape[1], x.shape[2], dy.shape[2]
        wd = ops.pack_conv3(list(ws), prec, tag, transpose=True)                     # [Z, Ci, 9*Cop]
        dx = ops.conv3x3(dy, wd, Ci, Co, B, H, W, prec, flip=1, dil=dil, out_dtype=xdt)
        conv = dict(H=H, W=W, C=Ci, Cp=Cip, dil=dil, flip=0)
        tiles = Z * (-(-Co // 128)) * (-(-9 * Cip // 128))
        S = _n_splits(tiles, B)                                          # slices = whole images (the gather decomposes pixel -> (y, x))
        p256 = Z * (-(-Co // 256)) * (-(-9 * Cip // 256))
        if (prec.name == "bf16" and dy.dtype == torch.bfloat16 and x.dtype == torch.bfloat16 and p256 >= 64
                and 100 * Co * 9 * Cip >= 45 * (-(-Co // 256)) * (-(-9 * Cip // 256)) * 65536):
            # the token-major kernel takes this launch (256 x 256 tiles, one per CU): slice the pixels so that the workgroups fill whole
            # rounds of the 256 CUs — InvPT's 576-channel convs at 128 x 128: 378 tiles = 1.48 rounds, two slices = 2.95
            best = None
            for cand in range(1, min(B, 8) + 1):
                if B % cand:
                    continue
                cost = -(-p256 * cand // N_CUS) / cand + 0.02 * (cand - 1)        # rounds per slice + the slab it writes and re-reads
                if best is None or cost < best[0] - 1e-9:
                    best = (cost, cand)
            S = best[1]
        while B % S:
            S -= 1
        if S == 1:
            dW = torch.empty(Z, Co, 9 * Cip, dtype=torch.float32, device=x.device)
            _gemm(dy, x, dW, Co, 9 * Cip, rows, prec, a_op=OP_R, b_op=OP_CONV_R, lda=Cop, ldb=Cip, ldd=9 * Cip, batch=Z,
                  a_zo=rows * Cop, b_zo=rows * Cip, d_zo=Co * 9 * Cip, conv=conv)
        else:
            c = rows // S
            slabs = torch.empty(Z, S, Co, 9 * Cip, dtype=torch.float32, device=x.device)
            _gemm(dy, x, slabs, Co, 9 * Cip, c, prec, a_op=OP_R, b_op=OP_CONV_R, lda=Cop, ldb=Cip, ldd=9 * Cip, batch=Z * S, batch_inner=S,
                  a_zo=rows * Cop, a_zi=c * Cop, b_zo=rows * Cip, b_zi=c * Cip, d_zo=S * Co * 9 * Cip, d_zi=Co * 9 * Cip, conv=conv)
            dW = slabs.sum(1)
        dws = ops.unpack_grads(dW, 'conv3', [(Co, Ci, 3, 3)] * Z,       # dW[z][co, tap*Cip + ci] -> W[co, ci, tap]
                               lambda src, flat, offs: [ops.segment(src, z * Co * 9 * Cip, flat, offs[z], (Co, 9, Ci), (9 * Cip, Cip, 1), (Ci * 9, 1, 9))
                                                        for z in range(Z)])
        dbs = list(ops.colsum_batched(dy, Co, Z, dy.stride(0)).unbind(0)) if has_bias else [None] * Z
        return (dx, None, None, None) + tuple(dws) + tuple(dbs)


class UpConv3x3Fn(Function):
    """F.interpolate(x, scale_factor=4, 'bilinear') -> task-batched Conv2d(3x3, padding 1) (+bias) on the LOW-resolution task stack
    x [Z, B*h*w, pitch(Ci)] (taskprompter.py:420 -> ConvHead.mt_proj[0], :692), "taps first" (include/mtt_hip.h, mtt_upconv_desc):
    forward = ONE GEMM with the nine stacked tap matrices + the expansion kernel; backward = the gather kernel + the input / weight
    gradient GEMMs of that linear layer on the h x w map.  geo = (B, h, w, Co, Ci)."""

    @staticmethod
    def forward(ctx, x, geo, prec, tag, *wb):
        B, h, w, Co, Ci = geo
        Z = len(wb) // 2
        ws, bs = wb[:Z], wb[Z:]
        xa = x if x.dtype == prec.adt else ops.cast_rows(x.reshape(-1, x.shape[-1]), prec.adt).view(x.shape)
        if prec.split and ops.split_gemm_ok(xa.shape[-1]):
            # x3f: planes of the low-resolution task features + pre-split tap matrices -> the split-plane LDS-DMA kernel; the hi planes
            # are the bf16 operands of the backward
            Zs, M, Kp = xa.shape
            sp = ops.split_cast(xa.reshape(Zs * M, Kp))
            xs = ops.Split(sp.hi.view(Zs, M, Kp), sp.lo.view(Zs, M, Kp))
            w9 = ops.pack_upconv9_split(list(ws), tag)
            y = ops.upconv3x3(xs, w9, Co, B, h, w, prec, bias=ops.stack_vec(list(bs), (tag, 'b')))
            ctx.save_for_backward(xs.hi, w9.hi)
        else:
            w9 = ops.pack_upconv9(list(ws), prec, tag)
            y = ops.upconv3x3(xa, w9, Co, B, h, w, prec, bias=ops.stack_vec(list(bs), (tag, 'b')))
            ctx.save_for_backward(xa, w9)
        ctx.meta = (geo, prec, Z, x.dtype)
        return y

    @staticmethod
    def backward(ctx, dy):
        xa, w9 = ctx.saved_tensors
        (B, h, w, Co, Ci), prec, Z, xdtype = ctx.meta
        prec = prec.bwd
        dy = dy.contiguous()
        M, Kp, N9 = xa.shape[1], xa.shape[2], w9.shape[1]
        dz = _to_bwd(ops.upconv4_gather(dy, Co, B, h, w), prec)          # [Z, M, N9]
        xa = _to_bwd(xa, prec)
        dx = torch.empty(Z, M, Kp, dtype=xdtype, device=dy.device)
        if prec.name == "bf16" and FAST_BWD and M >= FAST_MIN_ROWS and Kp >= 128:
            wT = w9.transpose(1, 2).contiguous().to(torch.bfloat16)      # [Z, Kp, N9]: reduction-contiguous dgrad operand (x3f: fp32 pack)
            _gemm(dz, wT, dx, M, Kp, N9, prec, lda=N9, ldb=N9, ldd=Kp, batch=Z, a_zo=M * N9, b_zo=Kp * N9, d_zo=M * Kp, n_store=Kp)
        else:
            _gemm(dz, w9, dx, M, Kp, N9, prec, b_op=OP_R, lda=N9, ldb=Kp, ldd=Kp, batch=Z, a_zo=M * N9, b_zo=N9 * Kp, d_zo=M * Kp,
                  n_store=Kp)
        dW9 = _wgrad_batched(dz, xa, N9, Kp, M, prec, Z, N9, Kp, M * N9, M * Kp)              # [Z, N9, Kp] fp32
        Cop = N9 // 9
        dws = ops.unpack_grads(dW9, 'upconv9', [(Co, Ci, 3, 3)] * Z,    # dW9[z][tap*Cop + co, ci] -> W[co, ci, tap]
                               lambda src, flat, offs: [ops.segment(src, z * N9 * Kp, flat, offs[z], (9, Co, Ci), (Cop * Kp, Kp, 1), (1, Ci * 9, 9))
                                                        for z in range(Z)])
        dbs = list(ops.colsum_batched(dy, Co, Z, dy.stride(0)).unbind(0))
        return (dx, None, None, None) + tuple(dws) + tuple(dbs)


class BnActStackFn(Function):
    """BatchNorm2d (+GELU / ReLU) over a task stack [Z, rows, ld], one BatchNorm holder per task (taskprompter.py:362,692,705;
    invpt.py:14).  training: centred batch statistics, merged across ranks in one collective when the holders are SyncBatchNorm
    (main.py:92), + running-stat update; eval: running statistics.  Every step is ONE Z-batched launch on slices of one input and
    one output tensor, so autograd never slices / re-stacks the multi-GB head maps."""

    @staticmethod
    def forward(ctx, x, C, act, training, bns, *gb):
        Z = x.shape[0]
        gammas, betas = torch.stack([g.detach() for g in gb[:Z]]), torch.stack([b.detach() for b in gb[Z:]])
        if training:
            y, mean, rstd, scale = bn_mod.train_forward(x, C, bns, act, gammas, betas)
        else:
            mean = torch.stack([bn.running_mean for bn in bns])
            eps = torch.tensor([bn.eps for bn in bns], dtype=torch.float32, device=x.device)[:, None] if len({bn.eps for bn in bns}) > 1 else bns[0].eps
            rstd = torch.rsqrt(torch.stack([bn.running_var for bn in bns]) + eps)
            y, scale = ops.bn_apply(x, C, mean, rstd, gammas, betas, act), 1.0
        ctx.save_for_backward(x, mean, rstd, gammas, betas)
        ctx.meta = (C, act, training, scale, bns)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, mean, rstd, gammas, betas = ctx.saved_tensors
        C, act, training, scale, bns = ctx.meta
        Z = x.shape[0]
        dy = dy.contiguous()
        s = ops.bn_bwd_reduce(x, dy, C, mean, rstd, gammas, betas, act)
        dgs, dbs = s[1].unbind(0), s[0].unbind(0)
        if training:
            red = bn_mod.sync_backward_sums(s.clone(), bns) * scale      # kernel divides by its local row count
        else:
            red = torch.zeros_like(s)
        dx = ops.bn_bwd_apply(x, dy, C, mean, rstd, gammas, betas, act, red)
        return (dx, None, None, None, None) + tuple(dgs) + tuple(dbs)


def bn_act_single(x, gamma, beta, bn, C, act, training):
    # (squeeze, not [0]: the backward of a select zero-fills and copies a whole map — 2.4 GB per head stage at the Swin-B benchmark shape)
    return BnActStackFn.apply(x[None], C, act, training, [bn], gamma, beta).squeeze(0)


class TaskHeadsFn(Function):
    """The per-task 1x1 prediction convs (different output widths) on a task stack y [Z, rows, ld]: returns Z fp32 maps
    [1, rows, pitch(n_z)].  One Function so that the backward writes each task's input gradient straight into its slice of a
    single dy buffer (taskprompter.py:694 linear_pred / transformer_decoder.py:130)."""

    @staticmethod
    def forward(ctx, y, prec, tag, *wb):
        Z = y.shape[0]
        ws, bs = wb[:Z], wb[Z:]
        outs, packs = [], []
        for z in range(Z):
            n = ws[z].shape[0]
            wp = ops.pack_linear([ws[z]], prec, (tag, z))
            outs.append(ops.linear(y[z], wp, n, prec, bias=bs[z][None], out_dtype=torch.float32))
            packs.append(wp)
        ctx.save_for_backward(y, *packs)
        ctx.meta = (prec, [tuple(w.shape) for w in ws])
        return tuple(outs)

    @staticmethod
    def backward(ctx, *dps):
        y, packs = ctx.saved_tensors[0], ctx.saved_tensors[1:]
        prec, wshapes = ctx.meta
        prec = prec.bwd
        Z, rows, ld = y.shape
        # ConvHeadFn (one node for conv -> BatchNorm -> predictions) stores this map in the backward's own dtype; as a node of its own,
        # autograd requires the dtype of y
        dy = torch.empty(y.shape, dtype=getattr(ctx, "grad_dtype", None) or y.dtype, device=y.device)
        dws, dbs = [], []
        for z in range(Z):
            n = wshapes[z][0]
            g = dps[z].contiguous().view(rows, -1)
            Kp = packs[z].shape[-1]
            ga = g                                                # the weight gradient's dy operand
            if (prec.name == "bf16" and FAST_BWD and HEAD_DGRAD_DMA and dy.dtype == torch.bfloat16 and rows >= FAST_MIN_ROWS and g.shape[1] % 8 == 0
                    and Kp == ld):
                # dya = g W is an outer-product-like GEMM (K = n <= 21 classes, a million rows): bound by the 0.7 GB it writes.  bf16 copies of
                # the two small operands (g [rows, pitch(n)], W^T [ld, pitch(n)]) put it on the 128-row LDS-DMA kernel instead of the
                # register-staged one (423 us per task at the benchmark's batch); padding columns of g are zeros (BilinearFn.backward)
                # only the n real columns are copied, the padding columns [n, pitch(n)) are written as zeros by the cast itself: the reduction
                # runs over pitch(n) columns and 0 * NaN is NaN, so the K padding must not depend on what the producer left there
                npad = g.shape[1]
                g16 = ops.cast2d(g, rows, n, g.stride(0), torch.bfloat16, ldd=npad, zero_pad=True)
                wT = _pad_last(packs[z][0].t(), npad).to(torch.bfloat16)                         # [ld, pitch(n)] (tiny)
                _gemm(g16, wT, dy[z], rows, ld, npad, prec, lda=npad, ldb=npad, ldd=ld, n_store=ld, variant=_lib.GEMM_DMA128)
                if y.dtype == torch.bfloat16:                     # (the activated map saved as bf16: ConvHeadFn's prologue form) both operands bf16
                    ga = g16
            else:
                _gemm(g, packs[z][0], dy[z], rows, min(ld, Kp), n, prec, b_op=OP_R, lda=g.shape[1], ldb=Kp, ldd=ld, n_store=ld)
            dW = _wgrad(ga, y[z], n, Kp, prec)
            dws.append(dW[:, :math.prod(wshapes[z][1:])].reshape(wshapes[z]))
            dbs.append(_colsum(g, n))
        return (dy, None, None) + tuple(dws) + tuple(dbs)


class _SubCtx:
    """ctx stand-in for running the forward / backward of a Function as a STAGE of a larger node (ConvHeadFn)."""

    def __init__(self, saved=()):
        self.saved_tensors = tuple(saved)

    def save_for_backward(self, *ts):
        self.saved_tensors = ts

    # what a stage may call on a real ctx and has no meaning inside the larger node (its outputs' flags are set by the outer forward,
    # every stage gradient is computed: the outer node owns needs_input_grad)
    needs_input_grad = (True,) * 64

    def mark_dirty(self, *ts):
        pass

    def mark_non_differentiable(self, *ts):
        pass

    def set_materialize_grads(self, value):
        pass


class ConvHeadFn(Function):
    """A conv head of ALL tasks as ONE autograd node: [up4 +] Conv3x3 -> BatchNorm (batch statistics) + activation -> the per-task 1x1
    predictions — ConvHead behind the x4 resize (taskprompter.py:420, :688-698: UpConv3x3Fn -> BnActStackFn -> TaskHeadsFn) and InvPT's
    mt_proj + MLPHead (invpt.py:538-541, transformer_decoder.py:124-131: Conv3x3Fn -> BnActStackFn -> TaskHeadsFn); the same kernels in the
    same order.  What the fusion of the NODE buys is the backward's storage: between separate nodes a gradient must have its tensor's
    dtype, and in the x3f mode (fp32-class forward on fp32-stored head maps, bf16 backward) that made every gradient map of the head —
    6 tasks x B x 128 x 128 x 352 fp32 = 8.7 GB at the benchmark's batch — an fp32 tensor that the bf16-arithmetic backward rounded while
    reading.  Inside one node the maps are stored in the backward's own dtype (`prec.bwd.adt`): the prediction dgrad writes bf16,
    BatchNorm's reduce / apply read and write bf16 next to the fp32 conv output (mtt_bn_desc.g_dtype), the gather kernel / conv dgrad
    read bf16 (no cast pass).  spec = (first stage 'up' | 'conv', its geo, its tag, C, act, prediction tag);
    params = Z conv weights, Z conv biases (or None), Z BN weights, Z BN biases, Z prediction weights, Z prediction biases."""

    @staticmethod
    def forward(ctx, fea, spec, prec, training, bns, *params):
        kind, geo, tag1, C, act, ptag = spec
        Z = len(params) // 6
        cw, cb, bg, bb, pw, pb = (params[i * Z:(i + 1) * Z] for i in range(6))
        c1, c2, c3 = _SubCtx(), _SubCtx(), _SubCtx()
        Stage1 = UpConv3x3Fn if kind == 'up' else Conv3x3Fn
        y = Stage1.forward(c1, fea, geo, prec, tag1, *cw, *cb)
        Zy, rows, ld = y.shape
        if (HEAD_PROLOGUE and training and prec.split and y.dtype == torch.float32
                and all(ops.head_prologue_ok(rows, ld, w.shape[0], ld) for w in pw)):
            # x3f training (round 6): BatchNorm + activation ride on the prediction GEMM's operand load (mtt_gemm_desc.a_scale): the activated
            # map is never written in fp32 (8.7 GB at the benchmark's batch) nor re-read — one pass over the conv output yields the fp32-class
            # predictions and the bf16 copy of the activated map that the bf16 backward needs for the predictions' weight gradients.
            # Same saved-tensor layout as the three-stage form, so the backward below is unchanged (it sees a bf16 `ya`).
            gammas, betas = torch.stack([g.detach() for g in bg]), torch.stack([b.detach() for b in bb])
            mean, rstd, scale = bn_mod.train_stats(y, C, bns)
            a_sc = torch.zeros(Zy, ld, dtype=torch.float32, device=y.device)
            a_sh = torch.zeros(Zy, ld, dtype=torch.float32, device=y.device)
            a_sc[:, :C] = rstd * gammas
            a_sh[:, :C] = betas - mean * a_sc[:, :C]
            ya16 = torch.empty(Zy, rows, ld, dtype=torch.bfloat16, device=y.device)
            preds, packs = [], []
            for z in range(Zy):
                n = pw[z].shape[0]
                wp = ops.pack_linear([pw[z]], prec, (ptag, z))
                preds.append(ops.linear(y[z], wp, n, prec, bias=pb[z][None], out_dtype=torch.float32, a_affine=(a_sc[z], a_sh[z], act, ya16[z])))
                packs.append(wp)
            preds = tuple(preds)
            c2.save_for_backward(y, mean, rstd, gammas, betas)
            c2.meta = (C, act, training, scale, bns)
            c3.save_for_backward(ya16, *packs)
            c3.meta = (prec, [tuple(w.shape) for w in pw])
        else:
            ya = BnActStackFn.forward(c2, y, C, act, training, bns, *bg, *bb)
            preds = TaskHeadsFn.forward(c3, ya, prec, ptag, *pw, *pb)
        n1, n2 = len(c1.saved_tensors), len(c2.saved_tensors)
        ctx.save_for_backward(*c1.saved_tensors, *c2.saved_tensors, *c3.saved_tensors)
        ctx.sub = (n1, n2, c1.meta, c2.meta, c3.meta, Z, Stage1)
        ctx.grad_dtype = prec.bwd.adt
        return preds

    @staticmethod
    def backward(ctx, *dps):
        n1, n2, m1, m2, m3, Z, Stage1 = ctx.sub
        sv = ctx.saved_tensors
        c1, c2, c3 = _SubCtx(sv[:n1]), _SubCtx(sv[n1:n1 + n2]), _SubCtx(sv[n1 + n2:])
        c1.meta, c2.meta, c3.meta = m1, m2, m3
        c3.grad_dtype = ctx.grad_dtype
        g3 = TaskHeadsFn.backward(c3, *dps)                 # (dya in grad_dtype, None, None, dW_pred x Z, db_pred x Z)
        g2 = BnActStackFn.backward(c2, g3[0])               # (dy in grad_dtype, None x 4, dgamma x Z, dbeta x Z)
        g1 = Stage1.backward(c1, g2[0])                     # (dfea in fea's dtype, None x 3, dW_conv x Z, db_conv x Z)
        return (g1[0], None, None, None, None) + tuple(g1[4:]) + tuple(g2[5:]) + tuple(g3[3:])


class FuseTailFn(Function):
    """fea_fuse[1..4] of all tasks at one tap as ONE autograd node (taskprompter.py:362, :472-474): Conv3x3 -> BatchNorm (batch statistics)
    + GELU -> Conv1x1 — Conv3x3Fn -> BnActStackFn -> BLinearFn, the same kernels in the same order, with what the single node allows:
    the gradient maps between the stages are stored in the backward's dtype (bf16 in x3f instead of fp32: no cast passes, half the
    traffic), and in x3f the activated map is handed to the 1x1 as hi / lo planes (one split pass; 880 -> 490 + 218 us per tap on the
    split-plane kernel at the benchmark's batch) whose hi plane is the bf16 operand the backward needs — the fp32 map is not kept.
    geo = Conv3x3Fn's geo (with the lo plane of a split input as 7th element); params = Z conv weights, Z conv biases, Z BN weights,
    Z BN biases, Z 1x1 weights, Z 1x1 biases."""

    @staticmethod
    def forward(ctx, y0, geo, tags, C, prec, training, bns, *params):
        Z = len(params) // 6
        cw, cb, bg, bb, w4, b4 = (params[i * Z:(i + 1) * Z] for i in range(6))
        c1, c2, c3 = _SubCtx(), _SubCtx(), _SubCtx()
        y1 = Conv3x3Fn.forward(c1, y0, geo, prec, tags[0], *cw, *cb)
        ya = BnActStackFn.forward(c2, y1, C, ACT_GELU, training, bns, *bg, *bb)
        if prec.split and ya.dtype == torch.float32 and ops.split_gemm_ok(ya.shape[-1]):
            Zs, M, Kp = ya.shape
            sp = ops.split_cast(ya.view(Zs * M, Kp))
            fea = BLinearFn.forward(c3, sp.hi.view(Zs, M, Kp), C, 'plain', None, None, prec, tags[1], sp.lo.view(Zs, M, Kp), *w4, *b4)
        else:
            fea = BLinearFn.forward(c3, ya, C, 'plain', None, None, prec, tags[1], None, *w4, *b4)
        n1, n2 = len(c1.saved_tensors), len(c2.saved_tensors)
        ctx.save_for_backward(*c1.saved_tensors, *c2.saved_tensors, *c3.saved_tensors)
        ctx.sub = (n1, n2, c1.meta, c2.meta, c3.meta, Z)
        ctx.grad_dtype = prec.bwd.adt
        return fea

    @staticmethod
    def backward(ctx, dfea):
        n1, n2, m1, m2, m3, Z = ctx.sub
        sv = ctx.saved_tensors
        c1, c2, c3 = _SubCtx(sv[:n1]), _SubCtx(sv[n1:n1 + n2]), _SubCtx(sv[n1 + n2:])
        c1.meta, c2.meta, c3.meta = m1, m2, m3
        c3.grad_dtype = ctx.grad_dtype
        g3 = BLinearFn.backward(c3, dfea)                   # (dya in grad_dtype, None x 7, dW4 x Z, db4 x Z)
        g2 = BnActStackFn.backward(c2, g3[0])               # (dy1 in grad_dtype, None x 4, dgamma x Z, dbeta x Z)
        g1 = Conv3x3Fn.backward(c1, g2[0])                  # (dy0 in y0's dtype, None x 3, dW_conv x Z, db_conv x Z)
        return (g1[0], None, None, None, None, None, None) + tuple(g1[4:]) + tuple(g2[5:]) + tuple(g3[8:])


class CtrMixFn(Function):
    """acc (+)= sum_s wmix[b,t,s] * fea[s]  — cross-task reweighting fused with the 4-tap sum (taskprompter.py:411,484)."""

    @staticmethod
    def forward(ctx, fea, wmix, acc, B, C):
        out = ops.ctr_mix(fea, wmix, B, C, acc)
        if acc is not None:
            ctx.mark_dirty(acc)
        ctx.save_for_backward(fea, wmix)
        ctx.meta = (B, C, acc is not None)
        return out

    @staticmethod
    def backward(ctx, dout):
        fea, wmix = ctx.saved_tensors
        B, C, had_acc = ctx.meta
        dout = dout.contiguous()
        T, rows, ld = fea.shape
        dfea = ops.ctr_mix(dout, wmix.transpose(1, 2).contiguous(), B, C, None, out_dtype=fea.dtype)   # written in the features' dtype
        dw = torch.empty_like(wmix)
        ops.call("ctr_dw", fea=fea, out=None, wmix=None, T=T, B=B, rows_per_b=rows // B, ld=ld, C=C, fea_dtype=dtype_code(fea),
                 accumulate=0, xargs=[dout, dw, ops.ws_for("ctr_dw", dout.device, T=T, B=B, rows_per_b=rows // B)])
        return dfea, dw, (dout if had_acc else None), None, None


class CtrWeightsFn(Function):
    """The [B, T, T] mixing weights of the cross-task reweighting: task t's two 1x1 convs over the head dimension (GELU between) applied to
    the prompt<->prompt raw logits (taskprompter.py:482-484) — mtt_ctr_weights / mtt_ctr_weights_bwd.  params = T first-conv weights
    [nH, nH, 1, 1], T biases [nH], T second-conv weights [1, nH, 1, 1], T biases [1]."""

    @staticmethod
    def forward(ctx, rawlog, B, T, tag, *params):
        w0s, b0s, w2s, b2s = (list(params[i * T:(i + 1) * T]) for i in range(4))
        W0, b0 = ops.stack_vec(w0s, (tag, 'w0')), ops.stack_vec(b0s, (tag, 'b0'))
        W2, b2 = ops.stack_vec(w2s, (tag, 'w2')), ops.stack_vec(b2s, (tag, 'b2'))
        rawlog = rawlog.contiguous()
        wmix = ops.ctr_weights(rawlog, W0, b0, W2, b2, B, T)
        ctx.save_for_backward(rawlog, W0, b0, W2, b2)
        ctx.meta = (B, T, [tuple(w.shape) for w in w0s], [tuple(w.shape) for w in w2s])
        return wmix

    @staticmethod
    def backward(ctx, dwmix):
        rawlog, W0, b0, W2, b2 = ctx.saved_tensors
        B, T, sh0, sh2 = ctx.meta
        nH, N = rawlog.shape[1], rawlog.shape[3]
        drawlog = torch.zeros_like(rawlog)                   # the kernel writes the first T columns; the patch columns get no gradient here
        dW0, db0, dW2, db2 = torch.empty_like(W0), torch.empty_like(b0), torch.empty_like(W2), torch.empty_like(b2)
        ops.call("ctr_weights_bwd", rawlog=rawlog, w0=W0, b0=b0, w2=W2, b2=b2, wmix=None, B=B, T=T, nH=nH, N=N,
                 xargs=[dwmix.contiguous(), drawlog, dW0, db0, dW2, db2])
        return ((drawlog, None, None, None) + tuple(dW0[t].view(sh0[t]) for t in range(T)) + tuple(db0.unbind(0))
                + tuple(dW2[t].view(sh2[t]) for t in range(T)) + tuple(db2[t].view(1) for t in range(T)))


class Deconv2x2Fn(Function):
    """ConvTranspose2d(k=2, s=2) = GEMM with a pixel-shuffle store (taskprompter.py:705).  x [B*H*W, Cip] -> [B*2H*2W, pitch(Co)]."""

    @staticmethod
    def forward(ctx, x, weight, bias, geo, prec, tag):
        B, H, W = geo
        Ci, Co = weight.shape[0], weight.shape[1]
        Wd = ops._cached((tag, prec.name, id(weight)), [weight],
                         lambda: ops.pack_matrix(weight.detach().permute(2, 3, 1, 0).reshape(4 * Co, Ci), prec)[None])
        y = ops.deconv2x2(x, Wd, Co, Ci, B, H, W, prec, bias4=bias.detach().repeat(4).contiguous())
        ctx.save_for_backward(x, Wd)
        ctx.meta = (geo, prec, Ci, Co)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, Wd = ctx.saved_tensors
        (B, H, W), prec, Ci, Co = ctx.meta
        prec = prec.bwd
        N4, N4p = 4 * Co, pitch(4 * Co)
        # pixel-unshuffle of the gradient (pure re-indexing): g4[(b,y,x), (dy*2+dx)*Co + co] = dy[b, 2y+dy, 2x+dx, co]
        g4 = torch.zeros(B * H * W, N4p, dtype=dy.dtype, device=dy.device)
        g4[:, :N4] = dy.view(B, H, 2, W, 2, -1)[..., :Co].permute(0, 1, 3, 2, 4, 5).reshape(B * H * W, N4)
        dW4 = _wgrad(g4, x, N4, x.shape[1], prec)                                       # [4*Co, Cip]
        dweight = dW4[:, :Ci].reshape(2, 2, Co, Ci).permute(3, 2, 0, 1).contiguous()
        dbias = _colsum(g4, N4).view(4, Co).sum(0)
        dx = _dgrad(g4, Wd[0], B * H * W, x.shape[1], N4, prec, x.dtype)
        return dx, dweight, dbias, None, None, None


class BilinearFn(Function):
    """F.interpolate(mode='bilinear', align_corners=False) on NHWC maps; nchw=True -> fp32 [B, C, Ho, Wo]."""

    @staticmethod
    def forward(ctx, x, geo, out_dtype, nchw):
        B, C, Hi, Wi, Ho, Wo = geo
        ctx.meta = (geo, nchw, x.shape, x.dtype)
        return ops.bilinear(x, B, C, Hi, Wi, Ho, Wo, out_dtype, nchw=nchw)

    @staticmethod
    def backward(ctx, dy):
        (B, C, Hi, Wi, Ho, Wo), nchw, xshape, xdtype = ctx.meta
        dy = dy.contiguous()
        Z, _, ld = xshape
        din = torch.zeros(xshape, dtype=torch.float32, device=dy.device)
        if nchw:
            ops.call("bilinear_bwd", **{"in": dy}, out=din, B=B, C=C, Hin=Hi, Win=Wi, Hout=Ho, Wout=Wo, ld_in=ld, ld_out=0,
                     in_dtype=F32, out_dtype=F32, out_nchw=1, accumulate=1)
        else:
            ops.call("bilinear_bwd", **{"in": dy}, out=din, B=Z * B, C=ld, Hin=Hi, Win=Wi, Hout=Ho, Wout=Wo, ld_in=ld, ld_out=ld,
                     in_dtype=dtype_code(dy), out_dtype=F32, out_nchw=0, accumulate=1)
        if xdtype != torch.float32:
            din = ops.cast2d(din.view(-1, ld), din.numel() // ld, ld, ld, xdtype, ldd=ld).view(xshape)
        return din, None, None, None


# =================================================================================================
def _drop_scales(model, blk, i, B, device):
    """The reference's 4 independent per-sample DropPath draws of a block (taskprompter.py:273-277), as the two
    [B, 2] (prompt rows, patch rows) scale tables of the attention and MLP residual epilogues."""
    rate = blk.drop_path_rate
    override = getattr(model, "_drop_override", None)       # tests inject the oracle's masks: [4, B] already mask / keep
    if not model.training or (rate <= 0.0 and override is None):
        return None, None
    if override is not None:
        d = override[i].to(device)
    else:
        keep = 1.0 - rate
        d = torch.bernoulli(torch.full((4, B), keep, device=device)) / keep
    return torch.stack([d[2], d[0]], 1).contiguous(), torch.stack([d[3], d[1]], 1).contiguous()


def _drop_tables(model, B, device):
    """_drop_scales of EVERY block from one Bernoulli draw per step ([depth, 4, B] instead of depth x [4, B]: the per-block form cost ~8 tiny
    launches per block, 0.4 ms per step at depth 24).  -> list of (rs_attn, rs_mlp) per block; (None, None) where the rate is 0 / in eval."""
    blocks = list(model.blocks)
    rates = [float(blk.drop_path_rate) for blk in blocks]
    if getattr(model, "_drop_override", None) is not None or not model.training or max(rates) <= 0.0:
        return [_drop_scales(model, blk, i, B, device) for i, blk in enumerate(blocks)]
    cache = getattr(model, "_drop_keep_cache", None)
    if cache is None or cache[0] != (tuple(rates), B, str(device)):
        keep = torch.tensor([1.0 - r for r in rates], dtype=torch.float32, device=device).view(-1, 1, 1)
        cache = model._drop_keep_cache = ((tuple(rates), B, str(device)), keep, keep.expand(len(rates), 4, B).contiguous())
    _, keep, probs = cache
    d = torch.bernoulli(probs) / keep                                           # [depth, 4, B]
    att = torch.stack([d[:, 2], d[:, 0]], 2)                                    # [depth, B, 2] (prompt rows, patch rows)
    mlp = torch.stack([d[:, 3], d[:, 1]], 2)
    return [(att[i], mlp[i]) if r > 0.0 else (None, None) for i, r in enumerate(rates)]


def _bn_act(y, bns, C, act, training):
    return BnActStackFn.apply(y, C, act, training, list(bns), *[bn.weight for bn in bns], *[bn.bias for bn in bns])


def _task_features(model, xsrc, rawlog, rawchan, il, B, acc):
    p, prec = model.p, model.prec
    names = p.TASKS.NAMES
    T, C = len(names), model.embed_dim
    h, w = model.resolution
    N = T + h * w
    tar, F = p.embed_dim, p.final_embed_dim
    tarp = pitch(tar)
    nwin = int(math.isqrt(model.chan_nheads))
    sp = model._decoder_split()          # x3f: modulate and the fea_decode epilogue write hi / lo planes, both GEMMs on the split-plane kernel
    mod = ModulateFn.apply(xsrc, rawlog, rawchan, (B, N, T, C, h, w, nwin), prec, sp)
    mod, mod_lo = mod if sp else (mod, None)
    dec_w, dec_b = [], []
    for t in names:
        dec_w += [model.fea_decode_spa[il][t][0].weight, model.fea_decode_chan[il][t][0].weight]
        dec_b += [model.fea_decode_spa[il][t][0].bias, model.fea_decode_chan[il][t][0].bias]
    cat = BLinearFn.apply(mod, tar, 'catpair', None, "split" if sp else None, prec, ('dec', il), mod_lo, *dec_w, *dec_b)
    cat, cat_lo = cat if sp else (cat, None)
    del mod, mod_lo
    ff = [model.fea_fuse[il][t] for t in names]
    kmap = (2 * tarp, [(0, 0, tar), (tarp, tar, tar)])
    spc = model._decoder_conv_split()    # ... and fea_fuse[0]'s epilogue writes y0 as planes for the implicit-GEMM 3x3 on the same kernel
    y0 = BLinearFn.apply(cat, F, 'plain', kmap, "split" if spc else (torch.float32 if sp else None), prec, ('f0', il), cat_lo,
                         *[m[0].weight for m in ff], *[m[0].bias for m in ff])
    y0, y0_lo = y0 if spc else (y0, None)
    del cat, cat_lo
    geo1 = (B, h, w, F, F) + ((1, y0_lo) if spc else ())
    if FUSE_HEAD_NODE:
        bns = [m[2] for m in ff]
        fea = FuseTailFn.apply(y0, geo1, (('f1', il), ('f4', il)), F, prec, model.training, bns, *[m[1].weight for m in ff],
                               *[m[1].bias for m in ff], *[bn.weight for bn in bns], *[bn.bias for bn in bns],
                               *[m[4].weight for m in ff], *[m[4].bias for m in ff])
    else:
        y1 = Conv3x3Fn.apply(y0, geo1, prec, ('f1', il), *[m[1].weight for m in ff], *[m[1].bias for m in ff])
        y1 = _bn_act(y1, [m[2] for m in ff], F, ACT_GELU, model.training)
        fea = BLinearFn.apply(y1, F, 'plain', None, None, prec, ('f4', il), None, *[m[4].weight for m in ff], *[m[4].bias for m in ff])
    wmix = model._ctr_weights(rawlog, il, B, T)
    return CtrMixFn.apply(fea, wmix, acc, B, F)


def backbone_forward(model, img, upsample=True):
    """Autograd twin of TaskPrompter._forward_nograd -> [T, B*4h*4w, pitch(F)] task features (upsample=False: the fp32 h x w sums
    [T, B*h*w, pitch(F)] before the x4 resize, for heads that fuse it)."""
    p, prec = model.p, model.prec
    B = img.shape[0]
    assert tuple(img.shape[-2:]) == tuple(model.patch_embed.img_size)
    h, w = model.resolution
    hw, T, C, nH = h * w, model.prompts_len, model.embed_dim, model.num_heads
    N = T + hw
    nwin = int(math.isqrt(model.chan_nheads))
    XT = PatchEmbedFn.apply(img, model.patch_embed.proj.weight, model.patch_embed.proj.bias, model.pos_embed,
                            model.task_prompts, (B, N, T, hw), prec)
    acc = None
    rawlog = rawchan = None
    drops = _drop_tables(model, B, img.device)
    for i, blk in enumerate(model.blocks):
        a = blk.attn
        rs_attn, rs_mlp = drops[i]
        XT2, rawlog, rawchan = AttnHalfFn.apply(XT, blk.norm1.weight, blk.norm1.bias, blk.norm1.eps, a.qkv.weight, a.qkv.bias,
                                                a.proj.weight, a.proj.bias, a.token_trans.weight, a.token_trans.bias,
                                                a.token_trans1.weight, a.token_trans1.bias, rs_attn,
                                                (B, N, nH, T, h, w, nwin, model._side_channels_used(i)), prec, ('blk', i))
        XT = MlpHalfFn.apply(XT2, blk.norm2.weight, blk.norm2.bias, blk.norm2.eps, blk.mlp.fc1.weight, blk.mlp.fc1.bias,
                             blk.mlp.fc2.weight, blk.mlp.fc2.bias, rs_mlp, (B, N, T), prec, ('blk', i))
        if (i + 1) in model.select_list:
            acc = _task_features(model, XT, rawlog, rawchan, model._tap_index(i), B, acc)
    xf = LayerNormFn.apply(XT, model.norm.weight, model.norm.bias, model.norm.eps, prec, torch.float32)
    acc = _task_features(model, xf, rawlog, rawchan, 3, B, acc)
    return upsample4(acc, B, h, w, prec) if upsample else acc


FUSE_HEAD_NODE = True      # ConvHead as one autograd node (ConvHeadFn); tests set it to False for the three-node form
HEAD_PROLOGUE = True       # ... whose x3f training forward applies BatchNorm + activation in the prediction GEMM's operand load (A/B: False)


def upsample4(acc, B, h, w, prec):
    return BilinearFn.apply(acc, (B, acc.shape[-1], h, w, 4 * h, 4 * w), prec.adt, False)


def heads_forward(kind, heads, fea, B, h4, w4, target, prec, training, lowres=False):
    """Autograd twin of taskprompter.run_heads: fea [Z, B*h4*w4, pitch(F)] (lowres: [Z, B*(h4/4)*(w4/4), pitch(F)], ConvHeads only) ->
    list of fp32 NCHW predictions."""
    F = heads[0].mt_proj[0].weight.shape[0]
    outs = []
    if kind == 'conv':
        tgt = target or (h4, w4)
        conv_w, conv_b = [hd.mt_proj[0].weight for hd in heads], [hd.mt_proj[0].bias for hd in heads]
        pred_w, pred_b = [hd.linear_pred.weight for hd in heads], [hd.linear_pred.bias for hd in heads]
        if lowres and FUSE_HEAD_NODE:
            bns = [hd.mt_proj[1] for hd in heads]
            preds = ConvHeadFn.apply(fea, ('up', (B, h4 // 4, w4 // 4, F, F), 'hc9', F, ACT_GELU, 'hp'), prec, training, bns,
                                     *conv_w, *conv_b, *[bn.weight for bn in bns], *[bn.bias for bn in bns], *pred_w, *pred_b)
        else:
            if lowres:
                y = UpConv3x3Fn.apply(fea, (B, h4 // 4, w4 // 4, F, F), prec, 'hc9', *conv_w, *conv_b)
            else:
                y = Conv3x3Fn.apply(fea, (B, h4, w4, F, F), prec, 'hc', *conv_w, *conv_b)
            y = _bn_act(y, [hd.mt_proj[1] for hd in heads], F, ACT_GELU, training)
            preds = TaskHeadsFn.apply(y, prec, 'hp', *pred_w, *pred_b)
        for hd, pred in zip(heads, preds):
            n_out = hd.linear_pred.weight.shape[0]
            outs.append(BilinearFn.apply(pred, (B, n_out, h4, w4, tgt[0], tgt[1]), torch.float32, True))
        return outs
    F2 = F // 2
    tgt = target or (2 * h4, 2 * w4)
    feas = fea.unbind(0)                  # one stack in the backward instead of a zero-filled [Z, ...] gradient per task
    for i, hd in enumerate(heads):
        y = Deconv2x2Fn.apply(feas[i], hd.mt_proj[0].weight, hd.mt_proj[0].bias, (B, h4, w4), prec, 'hd0')
        y = bn_act_single(y, hd.mt_proj[1].weight, hd.mt_proj[1].bias, hd.mt_proj[1], F2, ACT_GELU, training)[None]
        y = Conv3x3Fn.apply(y, (B, 2 * h4, 2 * w4, F2, F2), prec, 'hd3', hd.mt_proj[3].weight, hd.mt_proj[3].bias)
        y = bn_act_single(y.squeeze(0), hd.mt_proj[4].weight, hd.mt_proj[4].bias, hd.mt_proj[4], F2, ACT_GELU, training)[None]
        n_out = hd.linear_pred.weight.shape[0]
        pred = BLinearFn.apply(y, n_out, 'plain', None, torch.float32, prec, 'hp', None, hd.linear_pred.weight, hd.linear_pred.bias)
        outs.append(BilinearFn.apply(pred, (B, n_out, 2 * h4, 2 * w4, tgt[0], tgt[1]), torch.float32, True))
    return outs
