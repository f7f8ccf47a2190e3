"""TEST INFRASTRUCTURE ONLY — CPU emulator of the libmtt_hip.so C ABI (include/mtt_hip.h).

`call(name, **fields)` has the signature of `multi-task-transformer_amd/_lib.call` but executes the
descriptor's CONTRACT with plain torch index arithmetic on the tensors' flat storages (fp64
accumulation).  It is independent of the kernels: it is itself checked against torch.nn.functional
ops in tests/test_abi_emul.py, then used (a) as the per-entry-point oracle of the `-m gpu` parity
tests and (b) monkeypatched over `_lib.call` by CPU tests to exercise the product's host-side wiring
(descriptor construction, strides, padding) without a GPU.  Never imported by the product.
"""
import math

import torch

F32, BF16, SPLIT = 0, 1, 2        # SPLIT: x = hi + lo, two bf16 planes (main pointer = hi, *_lo = lo)
OP_K, OP_R, OP_CONV_K, OP_CONV_R = 0, 1, 2, 3
ACT_NONE, ACT_GELU, ACT_RELU, ACT_GELU_BWD, ACT_RELU_BWD, ACT_GELU_DAUX, ACT_MUL_AUX = 0, 1, 2, 3, 4, 5, 6

ROUND_BF16_OPERANDS = True   # emulate the f32->bf16 operand rounding of MTT_PREC_BF16


def flat(t):
    """(1-D view of the whole storage, element offset of t's first element)."""
    f = torch.empty(0, dtype=t.dtype, device=t.device).set_(t.untyped_storage())
    return f, t.storage_offset()


def _rd(t, idx, valid=None):
    f, o = flat(t)
    idx = idx + o
    if valid is not None:
        idx = torch.where(valid, idx, torch.zeros_like(idx))
    v = f[idx].double()
    if valid is not None:
        v = torch.where(valid, v, torch.zeros_like(v))
    return v


def _wr(t, idx, val):
    f, o = flat(t)
    f[idx + o] = val.to(t.dtype)


def _rowoff(m, mb, bs, ld):
    if mb and mb > 0:
        return (m // mb) * bs + (m % mb) * ld
    return m * ld


def _gelu(x):
    return 0.5 * x * (1.0 + torch.erf(x / math.sqrt(2.0)))


def _gelu_grad(x):
    return 0.5 * (1.0 + torch.erf(x / math.sqrt(2.0))) + x * torch.exp(-0.5 * x * x) / math.sqrt(2.0 * math.pi)


def _bf16_round(v):
    return v.float().bfloat16().double()


def gemm(**kw):
    g = lambda k, dflt=0: kw.get(k, dflt) if kw.get(k, dflt) is not None else dflt
    A, B, D = kw["A"], kw["B"], kw["D"]
    M, N, K = kw["M"], kw["N"], kw["K"]
    a_op, b_op = g("a_op"), g("b_op")
    lda, ldb, ldd = g("lda"), g("ldb"), g("ldd")
    batch, bi = max(1, g("batch", 1)), max(1, g("batch_inner", 1))
    conv = kw.get("conv") or {}
    prec = g("prec")
    alpha = kw.get("alpha", 1.0)
    m = torch.arange(M)[:, None]
    n = torch.arange(N)[:, None]
    k = torch.arange(K)[None, :]
    a_split, b_split, d_split = g("a_dtype") == SPLIT, g("b_dtype") == SPLIT, g("d_dtype") == SPLIT
    if a_split or b_split:
        assert a_split and b_split and prec == 1 and b_op == OP_K, "split operands: x3, both split, B reduction-contiguous"
        if a_op == OP_K:
            assert K % 32 == 0 and K >= 64, "split operands: K % 32 == 0, K >= 64"
        else:                                             # variant 9: implicit-GEMM 3x3 on planes
            assert a_op == OP_CONV_K and conv["Cp"] % 32 == 0 and K == 9 * conv["Cp"] and not g("a_mb"), "split conv: channel pitch % 32 == 0"

    def conv_taps(idx_tap, flip):
        ty, tx = idx_tap // 3, idx_tap % 3
        if flip:
            ty, tx = 2 - ty, 2 - tx
        return (ty - 1) * conv["dil"], (tx - 1) * conv["dil"]

    for z in range(batch):
        zo, zi = divmod(z, bi)
        za = zo * g("a_zo") + zi * g("a_zi")
        zb = zo * g("b_zo") + zi * g("b_zi")
        zd = zo * g("d_zo") + zi * g("d_zi")
        # ---- A [M, K] ----
        if a_op == OP_K:
            Am = _rd(A, za + _rowoff(m, g("a_mb"), g("a_bs"), lda) + k)
            if a_split:
                Am = Am + _rd(kw["A_lo"], za + _rowoff(m, g("a_mb"), g("a_bs"), lda) + k)
        elif a_op == OP_R:
            Am = _rd(A, za + k * lda + m)
        elif a_op == OP_CONV_K:
            H, W, Cc, Cp = conv["H"], conv["W"], conv["C"], conv["Cp"]
            tap, ci = k // Cp, k % Cp
            dy, dx = conv_taps(tap, conv.get("flip", 0))
            y, x = (m // W) % H, m % W
            ok = (y + dy >= 0) & (y + dy < H) & (x + dx >= 0) & (x + dx < W) & (ci < Cc)
            Am = _rd(A, za + (m + dy * W + dx) * lda + ci, ok)
            if a_split:
                Am = Am + _rd(kw["A_lo"], za + (m + dy * W + dx) * lda + ci, ok)
        else:
            raise ValueError("bad a_op")
        # ---- B [N, K] ----
        if b_op == OP_K:
            Bm = _rd(B, zb + n * ldb + k)
            if b_split:
                Bm = Bm + _rd(kw["B_lo"], zb + n * ldb + k)
        elif b_op == OP_R:
            Bm = _rd(B, zb + k * ldb + n)
        elif b_op == OP_CONV_R:
            H, W, Cc, Cp = conv["H"], conv["W"], conv["C"], conv["Cp"]
            tap, ci = n // Cp, n % Cp
            dy, dx = conv_taps(tap, 0)
            y, x = (k // W) % H, k % W
            ok = (y + dy >= 0) & (y + dy < H) & (x + dx >= 0) & (x + dx < W) & (ci < Cc)
            Bm = _rd(B, zb + (k + dy * W + dx) * ldb + ci, ok)
        else:
            raise ValueError("bad b_op")
        if kw.get("a_scale") is not None:                           # A prologue (variant 11 only): act(A * scale[k] + shift[k]), bf16 side copy
            assert a_op == OP_K and batch == 1 and prec == 1 and N <= 32, "A prologue: the exact-fp32 tall GEMM only"
            kk = torch.arange(K)
            Am = _act(Am * _rd(kw["a_scale"], kk)[None, :] + _rd(kw["a_shift"], kk)[None, :], g("a_act"))
            if kw.get("a_aux16") is not None:
                _wr(kw["a_aux16"], m * g("ld_a16") + k, Am)
        if prec == 0 and ROUND_BF16_OPERANDS:
            Am, Bm = _bf16_round(Am), _bf16_round(Bm)
        v = alpha * (Am @ Bm.T)                                       # [M, N] fp64
        # ---- epilogue ----
        zc = zo * g("col_zo") + zi * g("col_zi")
        ncol = torch.arange(N)
        if kw.get("colscale") is not None:
            v = v * _rd(kw["colscale"], zc + ncol)[None, :]
        if kw.get("colshift") is not None:
            v = v + _rd(kw["colshift"], zc + ncol)[None, :]
        act = g("act")
        d_mb, d_bs = g("d_mb"), g("d_bs")
        mrow = torch.arange(M)
        auxrow = ((mrow // d_mb) * d_mb + (mrow % d_mb)) if d_mb and d_mb > 0 else mrow
        zaux = zo * g("aux_zo") + zi * g("aux_zi")
        aux_idx = zaux + auxrow[:, None] * g("ldaux") + ncol[None, :]
        if act == ACT_GELU_DAUX:                                     # D = GELU(z), aux_out = GELU'(z)
            _wr(kw["aux_out"], aux_idx, _gelu_grad(v))
            v = _gelu(v)
        elif kw.get("aux_out") is not None:
            _wr(kw["aux_out"], aux_idx, v)
        if act == ACT_GELU:
            v = _gelu(v)
        elif act == ACT_MUL_AUX:
            v = v * _rd(kw["aux_in"], aux_idx)
        elif act == ACT_RELU:
            v = torch.clamp_min(v, 0.0)
        elif act == ACT_GELU_BWD:
            v = v * _gelu_grad(_rd(kw["aux_in"], aux_idx))
        elif act == ACT_RELU_BWD:
            v = torch.where(_rd(kw["aux_in"], aux_idx) > 0, v, torch.zeros_like(v))
        if kw.get("rowscale") is not None:
            q = (mrow // d_mb) if d_mb and d_mb > 0 else torch.zeros_like(mrow)
            rem = (mrow % d_mb) if d_mb and d_mb > 0 else mrow
            rs = _rd(kw["rowscale"], q * 2 + (rem >= g("n_prompt")).long())
            v = v * rs[:, None]
        if kw.get("resid") is not None:
            zr = zo * g("r_zo") + zi * g("r_zi")
            ridx = zr + _rowoff(mrow, g("r_mb"), g("r_bs"), g("ldr"))[:, None] + ncol[None, :]
            v = v + _rd(kw["resid"], ridx)
        if g("store_mode") == 1:
            Hs, Ws, Co = kw["ps_H"], kw["ps_W"], kw["ps_Co"]
            q, co = ncol // Co, ncol % Co
            x, y, bb = mrow % Ws, (mrow // Ws) % Hs, mrow // (Ws * Hs)
            orow = (bb[:, None] * (2 * Hs) + 2 * y[:, None] + (q // 2)[None, :]) * (2 * Ws) + 2 * x[:, None] + (q % 2)[None, :]
            _wr(D, zd + orow * ldd + co[None, :], v)
        else:
            didx = zd + _rowoff(mrow, d_mb, d_bs, ldd)[:, None] + ncol[None, :]
            _wr(D, didx, v)
            if d_split:
                _wr(kw["D_lo"], didx, v - _bf16_round(v))
            if kw.get("colsum_out") is not None:         # column sums of the values AS STORED (bf16 D: rounded); batch == 1 only
                assert batch == 1 and kw.get("colsum_ws") is not None
                stored = _bf16_round(v) if D.dtype == torch.bfloat16 and not d_split else v
                _wr(kw["colsum_out"], ncol, stored.sum(0))
            n_store = g("n_store")
            if n_store and n_store > N:
                pad = torch.arange(N, n_store)
                for Dp in ([D, kw["D_lo"]] if d_split else [D]):
                    _wr(Dp, zd + _rowoff(mrow, d_mb, d_bs, ldd)[:, None] + pad[None, :], torch.zeros(M, n_store - N, dtype=torch.float64))


def attn_fwd(**kw):
    qkv, out = kw["qkv"], kw["out"]
    B, N, nH, T = kw["B"], kw["N"], kw["nH"], kw["T"]
    C = nH * 64
    f, o = flat(qkv)
    x = f[o:o + B * N * 3 * C].double().view(B, N, 3, nH, 64)
    split = kw.get("dtype") == SPLIT
    if split:
        fl, ol = flat(kw["qkv_lo"])
        x = x + fl[ol:ol + B * N * 3 * C].double().view(B, N, 3, nH, 64)
    if kw.get("prec", 0) == 0 and ROUND_BF16_OPERANDS:
        x = _bf16_round(x)
    q, k, v = (x[:, :, i].transpose(1, 2) for i in range(3))
    raw = q @ k.transpose(-1, -2)
    if kw.get("rawlog") is not None and T > 0:
        _wr(kw["rawlog"], torch.arange(B * nH * T * N), raw[:, :, :T, :].reshape(-1))
    s = raw * kw["scale"]
    if kw.get("lse") is not None:
        _wr(kw["lse"], torch.arange(B * nH * N), torch.logsumexp(s, dim=-1).reshape(-1))
    y = (torch.softmax(s, dim=-1) @ v).transpose(1, 2).reshape(-1)
    _wr(out, torch.arange(B * N * C), y)
    if split:
        _wr(kw["out_lo"], torch.arange(B * N * C), y - _bf16_round(y))


def attn_bwd(**kw):
    """xargs = [dout, drawlog | None, dqkv, stat [B,nH,2,pad4(N)]]; forward recomputed in fp64 from (bf16-rounded) qkv."""
    dout, drawlog, dqkv, dsum = kw["xargs"]
    B, N, nH, T = kw["B"], kw["N"], kw["nH"], kw["T"]
    C = nH * 64
    x = _rd(kw["qkv"], torch.arange(B * N * 3 * C)).view(B, N, 3, nH, 64)
    q, k, v = (x[:, :, i].transpose(1, 2) for i in range(3))                      # [B, nH, N, 64]
    g = _rd(dout, torch.arange(B * N * C)).view(B, N, nH, 64).transpose(1, 2)
    raw = q @ k.transpose(-1, -2)
    P = torch.softmax(raw * kw["scale"], dim=-1)
    dV = P.transpose(-1, -2) @ g
    dP = g @ v.transpose(-1, -2)
    D = (dP * P).sum(-1, keepdim=True)
    dS = kw["scale"] * P * (dP - D)
    if drawlog is not None and T > 0:
        dS[:, :, :T] += _rd(drawlog, torch.arange(B * nH * T * N)).view(B, nH, T, N)
    dQ, dK = dS @ k, dS.transpose(-1, -2) @ q
    out = torch.stack([dQ, dK, dV], 0).permute(1, 3, 0, 2, 4).reshape(-1)      # [B, N, 3, nH, 64]
    _wr(dqkv, torch.arange(B * N * 3 * C), out)
    Np = (N + 3) // 4 * 4
    stat = torch.zeros(B, nH, 2, Np, dtype=torch.float64)
    stat[:, :, 0, :N] = D[..., 0]
    stat[:, :, 1, :N] = _rd(kw["lse"], torch.arange(B * nH * N)).view(B, nH, N) * 1.4426950408889634
    _wr(dsum, torch.arange(stat.numel()), stat.reshape(-1))


def softmax_fwd(**kw):
    rows, cols, ld = kw["rows"], kw["cols"], kw["ld"]
    idx = torch.arange(rows)[:, None] * ld + torch.arange(cols)[None, :]
    s = _rd(kw["S"], idx) * kw["scale"]
    _wr(kw["P"], idx, torch.softmax(s, dim=-1))
    if ld > cols:
        pad = torch.arange(rows)[:, None] * ld + torch.arange(cols, ld)[None, :]
        _wr(kw["P"], pad, torch.zeros(rows, ld - cols, dtype=torch.float64))


def softmax_bwd(**kw):
    rows, cols, ld = kw["rows"], kw["cols"], kw["ld"]
    idx = torch.arange(rows)[:, None] * ld + torch.arange(cols)[None, :]
    P, dP = _rd(kw["P"], idx), _rd(kw["dP"], idx)
    dS = kw["scale"] * P * (dP - (dP * P).sum(-1, keepdim=True))
    if kw.get("extra") is not None:
        rpm, er, eld = kw["rows_per_mat"], kw["extra_rows"], kw["extra_ld"]
        r = torch.arange(rows)
        mat, rin = r // rpm, r % rpm
        sel = rin < er
        eidx = (mat[sel] * er + rin[sel])[:, None] * eld + torch.arange(cols)[None, :]
        dS[sel] = dS[sel] + _rd(kw["extra"], eidx)
    _wr(kw["dS"], idx, dS)
    if ld > cols:
        pad = torch.arange(rows)[:, None] * ld + torch.arange(cols, ld)[None, :]
        _wr(kw["dS"], pad, torch.zeros(rows, ld - cols, dtype=torch.float64))


def layernorm_fwd(**kw):
    rows, Cn = kw["rows"], kw["C"]
    r, c = torch.arange(rows)[:, None], torch.arange(Cn)[None, :]
    x = _rd(kw["x"], r * kw["ldx"] + c)
    mean = x.mean(-1, keepdim=True)
    var = ((x - mean) ** 2).mean(-1, keepdim=True)
    rstd = torch.rsqrt(var + kw["eps"])
    y = (x - mean) * rstd * _rd(kw["gamma"], c) + _rd(kw["beta"], c)
    _wr(kw["y"], r * kw["ldy"] + c, y)
    if kw.get("y_dtype") == SPLIT:
        _wr(kw["y_lo"], r * kw["ldy"] + c, y - _bf16_round(y))
    if kw.get("y32") is not None:
        _wr(kw["y32"], r * kw["ldy32"] + c, y)
    if kw.get("mean") is not None:
        _wr(kw["mean"], torch.arange(rows), mean[:, 0])
    if kw.get("rstd") is not None:
        _wr(kw["rstd"], torch.arange(rows), rstd[:, 0])


def layernorm_bwd(**kw):
    rows, Cn = kw["rows"], kw["C"]
    r, c = torch.arange(rows)[:, None], torch.arange(Cn)[None, :]
    x = _rd(kw["x"], r * kw["ldx"] + c)
    dy = _rd(kw["dy"], r * kw["ldy"] + c)
    mean = _rd(kw["mean"], torch.arange(rows))[:, None]
    rstd = _rd(kw["rstd"], torch.arange(rows))[:, None]
    xh = (x - mean) * rstd
    gm = dy * _rd(kw["gamma"], c)
    if kw.get("dx") is not None:
        dx = rstd * (gm - gm.mean(-1, keepdim=True) - xh * (gm * xh).mean(-1, keepdim=True))
        idx = r * kw["ldx"] + c
        _wr(kw["dx"], idx, _rd(kw["dx_in"] if kw.get("dx_in") is not None else kw["dx"], idx) + dx)
    if kw.get("dgamma") is not None:
        ci = torch.arange(Cn)
        _wr(kw["dgamma"], ci, (dy * xh).sum(0))
        _wr(kw["dbeta"], ci, dy.sum(0))


def patchify16(args):
    img, cols, B, H, W, odt = args[:6]
    h, w = H // 16, W // 16
    x = img.double().reshape(B, 3, h, 16, w, 16).permute(0, 2, 4, 1, 3, 5).reshape(B * h * w * 768)
    _wr(cols, torch.arange(B * h * w * 768), x)


def chan_logits(**kw):
    B, T, N, Cn, h, w, nh, nw = (kw[k] for k in ("B", "T", "N", "C", "h", "w", "nh", "nw"))
    f, o = flat(kw["xn"])
    xn = f[o:o + B * N * Cn].double().view(B, N, Cn)[:, T:]
    if kw.get("dtype") == 2:                                          # MTT_SPLIT: hi + lo planes of the normalised tokens (q fp32)
        fl, ol = flat(kw["xn_lo"])
        xn = xn + fl[ol:ol + B * N * Cn].double().view(B, N, Cn)[:, T:]
    qi = (torch.arange(B)[:, None, None] * T + torch.arange(T)[None, :, None]) * kw["ldq"] + torch.arange(h * w)[None, None, :]
    q = _rd(kw["q"], qi)                                              # [B,T,hw]
    wh, ww = h // nh, w // nw
    qw = q.view(B, T, nh, wh, nw, ww)
    xw = xn.reshape(B, nh, wh, nw, ww, Cn)
    r = torch.einsum("btiajk,biajkc->btijc", qw, xw).reshape(-1)      # [B,T,nwin,C]
    idx = torch.arange(B * T * nh * nw * Cn)
    _wr(kw["rawchan"], idx, r)                                        # written (ABI 6: deterministic partials, no atomics)


def modulate(**kw):
    B, T, N, Cn, h, w, nh, nw = (kw[k] for k in ("B", "T", "N", "C", "h", "w", "nh", "nw"))
    hg = kw.get("hg") or 64
    hw, nH, nwin = h * w, Cn // hg, nh * nw
    xi = torch.arange(B)[:, None, None] * kw["x_bs"] + torch.arange(hw)[None, :, None] * kw["x_ld"] + torch.arange(Cn)[None, None, :]
    x = _rd(kw["x"], xi)
    f, o = flat(kw["rawlog"])
    rl = f[o:o + B * nH * T * N].double().view(B, nH, T, N)
    f, o = flat(kw["rawchan"])
    rc = f[o:o + B * T * nwin * Cn].double().view(B, T, nwin, Cn)
    yy, xx = torch.arange(hw) // w, torch.arange(hw) % w
    win = (yy // (h // nh)) * nw + (xx // (w // nw))
    outs = []
    for t in range(T):
        a = rl[:, :, t, T:].transpose(1, 2).repeat_interleave(hg, dim=2)     # [B,hw,C]
        outs.append(x * (1 + a))
        outs.append(x * (1 + rc[:, t][:, win]))
    y = torch.stack(outs, 0).reshape(-1)
    _wr(kw["out"], torch.arange(y.numel()), y)
    if kw.get("out_lo") is not None:                 # MTT_SPLIT: out is the bf16 hi plane (rounded by the store), out_lo = bf16(v - hi)
        _wr(kw["out_lo"], torch.arange(y.numel()), y - _bf16_round(y))


def ctr_mix(**kw):
    T, B, rpb, ld, Cn = kw["T"], kw["B"], kw["rows_per_b"], kw["ld"], kw["C"]
    rows = B * rpb
    C8 = (Cn + 7) // 8 * 8
    idx = (torch.arange(T)[:, None, None] * rows + torch.arange(rows)[None, :, None]) * ld + torch.arange(C8)[None, None, :]
    fea = _rd(kw["fea"], idx).view(T, B, rpb, C8)
    wm = _rd(kw["wmix"], torch.arange(B * T * T)).view(B, T, T)
    out = torch.einsum("bts,sbrc->tbrc", wm, fea).reshape(T, rows, C8)
    if kw.get("accumulate"):
        assert kw.get("out_dtype", F32) == F32, "accumulating ctr_mix output is fp32"
        out = out + _rd(kw["out"], idx)
    assert kw.get("out_dtype", F32) == (BF16 if kw["out"].dtype == torch.bfloat16 else F32)
    _wr(kw["out"], idx, out)


def _ctrw_views(kw):
    B, T, nH, N = kw["B"], kw["T"], kw["nH"], kw["N"]
    idx = ((torch.arange(B)[:, None, None, None] * nH + torch.arange(nH)[None, :, None, None]) * T + torch.arange(T)[None, None, :, None]) * N \
        + torch.arange(T)[None, None, None, :]
    z = _rd(kw["rawlog"], idx).permute(0, 2, 3, 1)                          # [B, t, s, nH]
    w0 = _rd(kw["w0"], torch.arange(T * nH * nH)).view(T, nH, nH)
    b0 = _rd(kw["b0"], torch.arange(T * nH)).view(T, nH)
    w2 = _rd(kw["w2"], torch.arange(T * nH)).view(T, nH)
    b2 = _rd(kw["b2"], torch.arange(T))
    return z, w0, b0, w2, b2, idx


def _ctrw_math(z, w0, b0, w2, b2):
    hid = _gelu(torch.einsum("btsh,tjh->btsj", z, w0) + b0[None, :, None, :])
    return torch.einsum("btsj,tj->bts", hid, w2) + b2[None, :, None]


def ctr_weights(**kw):
    z, w0, b0, w2, b2, _ = _ctrw_views(kw)
    out = _ctrw_math(z, w0, b0, w2, b2)
    _wr(kw["wmix"], torch.arange(out.numel()), out.reshape(-1))


def ctr_weights_bwd(**kw):
    dwmix, drawlog, dw0, db0, dw2, db2 = kw["xargs"]
    z, w0, b0, w2, b2, idx = _ctrw_views(kw)
    g = _rd(dwmix, torch.arange(z.shape[0] * z.shape[1] * z.shape[2])).view(z.shape[:3])
    with torch.enable_grad():
        leaves = [t.clone().requires_grad_(True) for t in (z, w0, b0, w2, b2)]
        grads = torch.autograd.grad((_ctrw_math(*leaves) * g).sum(), leaves)
    _wr(drawlog, idx, grads[0].permute(0, 3, 1, 2))                           # only the first T columns are written
    for dst, gr in zip((dw0, db0, dw2, db2), grads[1:]):
        _wr(dst, torch.arange(gr.numel()), gr.reshape(-1))


def _detloss_elem(x, kw):
    """element-wise loss of mtt_detloss_desc as a differentiable fp64 expression (det_losses.py:102-123, :183-224)"""
    N, Cn = kw["N"], kw["C"]
    if kw["kind"] == 0:
        tgt = flat(kw["target"])[0][kw["target"].storage_offset():kw["target"].storage_offset() + N].long()
        onehot = torch.nn.functional.one_hot(tgt, Cn + 1)[:, :Cn].double()
        p = torch.sigmoid(x)
        pt = (1 - p) * onehot + p * (1 - onehot)
        fw = (kw["alpha"] * onehot + (1 - kw["alpha"]) * (1 - onehot)) * pt.pow(kw["gamma"])
        loss = torch.nn.functional.binary_cross_entropy_with_logits(x, onehot, reduction="none") * fw
    else:
        y = _rd(kw["target"], torch.arange(N * Cn)).view(N, Cn)
        d = (x - y).abs()
        loss = torch.where(d < kw["beta"], 0.5 * d * d / kw["beta"], d - 0.5 * kw["beta"])
    if kw.get("wmode", 0) == 1:
        loss = loss * _rd(kw["weight"], torch.arange(N)).view(N, 1)
    elif kw.get("wmode", 0) == 2:
        loss = loss * _rd(kw["weight"], torch.arange(N * Cn)).view(N, Cn)
    return loss


def detloss_fwd(**kw):
    N, Cn = kw["N"], kw["C"]
    x = _rd(kw["pred"], torch.arange(N * Cn)).view(N, Cn)
    loss = _detloss_elem(x, kw)
    if kw.get("out") is not None:
        _wr(kw["out"], torch.arange(N * Cn), loss.reshape(-1))
    if kw.get("sum") is not None:
        _wr(kw["sum"], torch.arange(1), loss.sum().reshape(1))


def detloss_bwd(**kw):
    gscale, gelem, scale, dpred = kw["xargs"]
    N, Cn = kw["N"], kw["C"]
    x = _rd(kw["pred"], torch.arange(N * Cn)).view(N, Cn)
    with torch.enable_grad():
        xr = x.clone().requires_grad_(True)
        up = _rd(gelem, torch.arange(N * Cn)).view(N, Cn) * scale if gelem is not None else _rd(gscale, torch.arange(1))[0] * scale
        (g,) = torch.autograd.grad((_detloss_elem(xr, kw) * up).sum(), xr)
    _wr(dpred, torch.arange(N * Cn), g.reshape(-1))


def _src(o, n_in, n_out):
    scale = torch.tensor(n_in / n_out, dtype=torch.float32)
    s = (torch.arange(n_out, dtype=torch.float32) + 0.5) * scale - 0.5
    s = torch.clamp_min(s, 0.0)
    i0 = torch.clamp_max(s.long(), n_in - 1)
    i1 = torch.where(i0 < n_in - 1, i0 + 1, i0)
    return i0, i1, (s - i0.float()).double()


def bilinear_fwd(**kw):
    B, Cn, Hi, Wi, Ho, Wo = (kw[k] for k in ("B", "C", "Hin", "Win", "Hout", "Wout"))
    if not kw.get("out_nchw"):
        Cn = (Cn + 7) // 8 * 8      # NHWC mode carries whole 8-channel chunks (zero padding in -> zero out)
    ii = (torch.arange(B * Hi * Wi)[:, None] * kw["ld_in"] + torch.arange(Cn)[None, :])
    x = _rd(kw["in"], ii).view(B, Hi, Wi, Cn)
    y0, y1, wy = _src(None, Hi, Ho)
    x0, x1, wx = _src(None, Wi, Wo)
    wy, wx = wy[None, :, None, None], wx[None, None, :, None]
    top = x[:, y0][:, :, x0] * (1 - wx) + x[:, y0][:, :, x1] * wx
    bot = x[:, y1][:, :, x0] * (1 - wx) + x[:, y1][:, :, x1] * wx
    y = top * (1 - wy) + bot * wy                                     # [B,Ho,Wo,C]
    if kw.get("out_nchw"):
        _wr(kw["out"], torch.arange(B * Cn * Ho * Wo), y.permute(0, 3, 1, 2).reshape(-1))
    else:
        oi = torch.arange(B * Ho * Wo)[:, None] * kw["ld_out"] + torch.arange(Cn)[None, :]
        v = y.reshape(B * Ho * Wo, Cn)
        if kw.get("accumulate"):
            v = v + _rd(kw["out"], oi)
        _wr(kw["out"], oi, v)


def bilinear_bwd(**kw):
    """in = dout (Hout x Wout), out = din fp32 NHWC (Hin x Win), accumulated."""
    B, Cn, Hi, Wi, Ho, Wo = (kw[k] for k in ("B", "C", "Hin", "Win", "Hout", "Wout"))
    if not kw.get("out_nchw"):
        Cn = (Cn + 7) // 8 * 8      # NHWC mode carries whole 8-channel chunks
    if kw.get("out_nchw"):
        f, o = flat(kw["in"])
        g = f[o:o + B * Cn * Ho * Wo].double().view(B, Cn, Ho, Wo).permute(0, 2, 3, 1)
    else:
        gi = torch.arange(B * Ho * Wo)[:, None] * kw["ld_out"] + torch.arange(Cn)[None, :]
        g = _rd(kw["in"], gi).view(B, Ho, Wo, Cn)
    y0, y1, wy = _src(None, Hi, Ho)
    x0, x1, wx = _src(None, Wi, Wo)
    din = torch.zeros(B, Hi, Wi, Cn, dtype=torch.float64)
    wyv, wxv = wy[None, :, None, None], wx[None, None, :, None]
    for ys, wyy in ((y0, 1 - wyv), (y1, wyv)):
        for xs, wxx in ((x0, 1 - wxv), (x1, wxv)):
            contrib = g * wyy * wxx
            tmp = torch.zeros(B, Hi, Wo, Cn, dtype=torch.float64).index_add_(1, ys, contrib)
            din.index_add_(2, xs, tmp)
    di = torch.arange(B * Hi * Wi)[:, None] * kw["ld_in"] + torch.arange(Cn)[None, :]
    _wr(kw["out"], di, _rd(kw["out"], di) + din.view(B * Hi * Wi, Cn))


def _act(u, act):
    return _gelu(u) if act == ACT_GELU else (torch.clamp_min(u, 0) if act == ACT_RELU else u)


def _act_grad(u, act):
    return _gelu_grad(u) if act == ACT_GELU else ((u > 0).double() if act == ACT_RELU else torch.ones_like(u))


def _bn_maps(kw):
    """(z, element index [rows, C] of map z, parameter index [C] of map z) for every map of a Z-batched BN descriptor."""
    Z = max(1, kw.get("Z", 1) or 1)
    base = torch.arange(kw["rows"])[:, None] * kw["ld"] + torch.arange(kw["C"])[None, :]
    for z in range(Z):
        yield z, base + z * (kw.get("x_zs", 0) or 0), torch.arange(kw["C"]) + z * (kw.get("p_zs", 0) or 0)


def _bn_pad(kw, t, idx):
    C8 = kw["ld"]                   # the whole pitch
    if C8 > kw["C"]:
        pad = idx[:, :1] + torch.arange(kw["C"], C8)[None, :]
        _wr(t, pad, torch.zeros(kw["rows"], C8 - kw["C"], dtype=torch.float64))


def bn_stats(**kw):
    for z, ix, pc in _bn_maps(kw):
        x = _rd(kw["x"], ix)
        mean = x.mean(0)
        _wr(kw["mean_out"], pc, mean)
        _wr(kw["m2_out"], pc, ((x - mean) ** 2).sum(0))


def bn_apply(**kw):
    for z, ix, pc in _bn_maps(kw):
        x = _rd(kw["x"], ix)
        u = (x - _rd(kw["mean"], pc)) * _rd(kw["rstd"], pc) * _rd(kw["gamma"], pc) + _rd(kw["beta"], pc)
        _wr(kw["y"], ix, _act(u, kw.get("act", 0)))
        _bn_pad(kw, kw["y"], ix)


def bn_bwd_reduce(**kw):
    for z, ix, pc in _bn_maps(kw):
        x, dy = _rd(kw["x"], ix), _rd(kw["dy"], ix)
        xh = (x - _rd(kw["mean"], pc)) * _rd(kw["rstd"], pc)
        du = dy * _act_grad(xh * _rd(kw["gamma"], pc) + _rd(kw["beta"], pc), kw.get("act", 0))
        _wr(kw["dsum"], pc, du.sum(0))
        _wr(kw["dsumxh"], pc, (du * xh).sum(0))


def bn_bwd_apply(**kw):
    n = kw["rows"]
    for z, ix, pc in _bn_maps(kw):
        x, dy = _rd(kw["x"], ix), _rd(kw["dy"], ix)
        rstd, gam = _rd(kw["rstd"], pc), _rd(kw["gamma"], pc)
        xh = (x - _rd(kw["mean"], pc)) * rstd
        du = dy * _act_grad(xh * gam + _rd(kw["beta"], pc), kw.get("act", 0))
        dx = gam * rstd * (du - _rd(kw["dsum"], pc) / n - xh * _rd(kw["dsumxh"], pc) / n)
        _wr(kw["dx"], ix, dx)
        _bn_pad(kw, kw["dx"], ix)


def cast2d(args):
    src, dst, rows, cols, lds, ldd, sdt, ddt, zp = args[:9]
    r, c = torch.arange(rows)[:, None], torch.arange(cols)[None, :]
    _wr(dst, r * ldd + c, _rd(src, r * lds + c))
    if zp and ldd > cols:
        _wr(dst, r * ldd + torch.arange(cols, ldd)[None, :], torch.zeros(rows, ldd - cols, dtype=torch.float64))


def pixshuf2(args):
    z, out, B, H, W, Co, ldz, ldo = args[:8]
    b, y, x, dy, dx, c = torch.meshgrid(torch.arange(B), torch.arange(H), torch.arange(W), torch.arange(2), torch.arange(2), torch.arange(Co), indexing="ij")
    src = ((b * H + y) * W + x) * ldz + (dy * 2 + dx) * Co + c
    opix = (b * 2 * H + 2 * y + dy) * (2 * W) + 2 * x + dx
    _wr(out, (opix * ldo + c).reshape(-1), _rd(z, src.reshape(-1)))
    if ldo > Co:
        pad = (opix[..., :1] * ldo + torch.arange(Co, ldo)).reshape(-1)
        _wr(out, pad, torch.zeros(pad.numel(), dtype=torch.float64))


def split_cast(args):
    src, hi, lo, rows, cols, lds, ldd = args[:7]
    r, c = torch.arange(rows)[:, None], torch.arange(cols)[None, :]
    x = _rd(src, r * lds + c)
    _wr(hi, r * ldd + c, x)
    _wr(lo, r * ldd + c, x - _bf16_round(x))
    if ldd > cols:
        for t in (hi, lo):
            _wr(t, r * ldd + torch.arange(cols, ldd)[None, :], torch.zeros(rows, ldd - cols, dtype=torch.float64))


def colsum(args):
    src, dst, rows, cols, ld, sdt = args[:6]
    r, c = torch.arange(rows)[:, None], torch.arange(cols)[None, :]
    ci = torch.arange(cols)
    _wr(dst, ci, _rd(src, r * ld + c).sum(0))


def colsum_batched(args):
    src, dst, rows, cols, ld, dt, Z, src_zs, dst_zs = args[:9]
    for z in range(Z):
        r, c = torch.arange(rows)[:, None], torch.arange(cols)[None, :]
        _wr(dst, z * dst_zs + torch.arange(cols), _rd(src, z * src_zs + r * ld + c).sum(0))


def add_rows(args):
    src, dst, rows, cols, lds, ldd, sdt, alpha = args[:8]
    r, c = torch.arange(rows)[:, None], torch.arange(cols)[None, :]
    _wr(dst, r * ldd + c, _rd(dst, r * ldd + c) + alpha * _rd(src, r * lds + c))



def modulate_bwd(**kw):
    dout, dx, drawlog, drawchan = kw["xargs"][:4]                     # [4] = workspace
    B, T, N, Cn, h, w, nh, nw = (kw[k] for k in ("B", "T", "N", "C", "h", "w", "nh", "nw"))
    hg = kw.get("hg") or 64
    hw, nH, nwin = h * w, Cn // hg, nh * nw
    xi = torch.arange(B)[:, None, None] * kw["x_bs"] + torch.arange(hw)[None, :, None] * kw["x_ld"] + torch.arange(Cn)[None, None, :]
    x = _rd(kw["x"], xi)
    f, o = flat(kw["rawlog"]); rl = f[o:o + B * nH * T * N].double().view(B, nH, T, N)
    f, o = flat(kw["rawchan"]); rc = f[o:o + B * T * nwin * Cn].double().view(B, T, nwin, Cn)
    f, o = flat(dout); g = f[o:o + 2 * T * B * hw * Cn].double().view(T, 2, B, hw, Cn)
    yy, xx = torch.arange(hw) // w, torch.arange(hw) % w
    win = (yy // (h // nh)) * nw + (xx // (w // nw))
    dxv = torch.zeros(B, hw, Cn, dtype=torch.float64)
    dl = torch.zeros(B, nH, T, N, dtype=torch.float64)
    dc = torch.zeros(B, T, nwin, Cn, dtype=torch.float64)
    for t in range(T):
        a = rl[:, :, t, T:].transpose(1, 2).repeat_interleave(hg, dim=2)
        dxv += g[t, 0] * (1 + a) + g[t, 1] * (1 + rc[:, t][:, win])
        dl[:, :, t, T:] = (g[t, 0] * x).view(B, hw, nH, hg).sum(-1).transpose(1, 2)
        dc[:, t].index_add_(1, win, g[t, 1] * x)
    _wr(dx, xi, dxv)                                                  # written (the rows outside the addressed block are the caller's)
    li = torch.arange(B * nH * T)[:, None] * N + torch.arange(T, N)[None, :]
    _wr(drawlog, li, dl.view(B * nH * T, N)[:, T:])
    ci = torch.arange(B * T * nwin * Cn)
    _wr(drawchan, ci, dc.reshape(-1))                                 # written


def chan_logits_bwd(**kw):
    drawchan, dq, dq_dtype, dxn = kw["xargs"]
    B, T, N, Cn, h, w, nh, nw = (kw[k] for k in ("B", "T", "N", "C", "h", "w", "nh", "nw"))
    hw, nwin = h * w, nh * nw
    f, o = flat(kw["xn"]); xn = f[o:o + B * N * Cn].double().view(B, N, Cn)[:, T:]
    qi = (torch.arange(B)[:, None, None] * T + torch.arange(T)[None, :, None]) * kw["ldq"] + torch.arange(hw)[None, None, :]
    q = _rd(kw["q"], qi)
    f, o = flat(drawchan); g = f[o:o + B * T * nwin * Cn].double().view(B, T, nwin, Cn)
    yy, xx = torch.arange(hw) // w, torch.arange(hw) % w
    win = (yy // (h // nh)) * nw + (xx // (w // nw))
    gp = g[:, :, win]                                                  # [B,T,hw,C]
    _wr(dq, qi, torch.einsum("btpc,bpc->btp", gp, xn))
    xi = (torch.arange(B)[:, None, None] * N + T + torch.arange(hw)[None, :, None]) * Cn + torch.arange(Cn)[None, None, :]
    _wr(dxn, xi, _rd(dxn, xi) + torch.einsum("btpc,btp->bpc", gp, q))


def ctr_dw(**kw):
    dout, dw = kw["xargs"][:2]                                        # [2] = workspace
    T, B, rpb, ld, Cn = kw["T"], kw["B"], kw["rows_per_b"], kw["ld"], kw["C"]
    rows = B * rpb
    C8 = (Cn + 7) // 8 * 8
    idx = (torch.arange(T)[:, None, None] * rows + torch.arange(rows)[None, :, None]) * ld + torch.arange(C8)[None, None, :]
    fea = _rd(kw["fea"], idx).view(T, B, rpb, C8)
    g = _rd(dout, idx).view(T, B, rpb, C8)
    wi = torch.arange(B * T * T)
    _wr(dw, wi, torch.einsum("tbrc,sbrc->bts", g, fea).reshape(-1))   # written


def rowscale_cast(args):
    src, dst, rows, cols, lds, ldd, sdt, ddt, rowscale, mb, n_prompt = args[:11]
    r, c = torch.arange(rows)[:, None], torch.arange(cols)[None, :]
    v = _rd(src, r * lds + c)
    if rowscale is not None:
        rr = torch.arange(rows)
        q = (rr // mb) if mb > 0 else torch.zeros_like(rr)
        rem = (rr % mb) if mb > 0 else rr
        v = v * _rd(rowscale, q * 2 + (rem >= n_prompt).long())[:, None]
    _wr(dst, r * ldd + c, v)
    return v


def rowscale_cast_colsum(args):
    """mtt_rowscale_cast_colsum: the cast + column sums of the values as stored in dst."""
    v = rowscale_cast(args[:11])
    dst, out = args[1], args[11]
    stored = _bf16_round(v) if dst.dtype == torch.bfloat16 else v
    _wr(out, torch.arange(args[3]), stored.sum(0))


def dwconv3x3s2(**kw):
    Z, B, H, W, ld = kw["Z"], kw["B"], kw["H"], kw["W"], kw["ld"]
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    x = _rd(kw["x"], torch.arange(Z * B * H * W * ld)).view(Z * B, H, W, ld).permute(0, 3, 1, 2)
    w = _rd(kw["w"], torch.arange(Z * 9 * ld)).view(Z, 3, 3, ld).permute(0, 3, 1, 2)            # [Z, ld, 3, 3]
    ys = []
    for z in range(Z):
        y = torch.nn.functional.conv2d(x[z * B:(z + 1) * B], w[z][:, None], None, stride=2, padding=1, groups=ld)
        if kw.get("scale") is not None:
            sc = _rd(kw["scale"], z * ld + torch.arange(ld)); sh = _rd(kw["shift"], z * ld + torch.arange(ld))
            y = y * sc[None, :, None, None] + sh[None, :, None, None]
        ys.append(y.permute(0, 2, 3, 1).reshape(-1))
    y = torch.cat(ys)
    _wr(kw["y"], torch.arange(y.numel()), y)


def avgpool_ceil(**kw):
    B, H, W, k, ld = kw["B"], kw["H"], kw["W"], kw["k"], kw["ld"]
    x = _rd(kw["x"], torch.arange(B * H * W * ld)).view(B, H, W, ld).permute(0, 3, 1, 2)
    y = torch.nn.functional.avg_pool2d(x, k, k, 0, ceil_mode=True).permute(0, 2, 3, 1).reshape(-1)
    _wr(kw["y"], torch.arange(y.numel()), y)


def layernorm_mt(**kw):
    rows, T, D, ldx, ldy = kw["rows"], kw["T"], kw["D"], kw["ldx"], kw["ldy"]
    idx = (torch.arange(T)[:, None, None] * rows + torch.arange(rows)[None, :, None]) * ldx + torch.arange(D)[None, None, :]
    x = _rd(kw["x"], idx)                                               # [T, rows, D]
    cat = x.permute(1, 0, 2).reshape(rows, T * D)
    mean = cat.mean(-1, keepdim=True); var = ((cat - mean) ** 2).mean(-1, keepdim=True)
    y = (cat - mean) * torch.rsqrt(var + kw["eps"]) * _rd(kw["gamma"], torch.arange(T * D)) + _rd(kw["beta"], torch.arange(T * D))
    y = y.view(rows, T, D).permute(1, 0, 2)
    full = torch.zeros(T, rows, ldy, dtype=torch.float64); full[..., :D] = y
    oi = (torch.arange(T)[:, None, None] * rows + torch.arange(rows)[None, :, None]) * ldy + torch.arange(ldy)[None, None, :]
    _wr(kw["y"], oi, full)


def attn_msg(**kw):
    B, heads, T, qh, qw, K, ldk, ldkp = (kw[k] for k in ("B", "heads", "T", "qh", "qw", "K", "ldk", "ldkp"))
    Q, sh, sw = T * qh * qw, qh // 2, qw // 2
    Qp = T * sh * sw
    ci = torch.arange(B * heads * Q)[:, None] * ldk + torch.arange(K)[None, :]
    cur = _rd(kw["cur"], ci).view(B, heads, Q, K)
    pi = torch.arange(B * heads * Qp)[:, None] * ldkp + torch.arange(K)[None, :]
    prev = _rd(kw["prev"], pi).view(B, heads, T, sh, sw, K)
    up = torch.nn.functional.interpolate(prev.permute(0, 1, 2, 5, 3, 4).reshape(B * heads * T, K, sh, sw), scale_factor=2,
                                         mode="bilinear", align_corners=False)
    up = up.view(B, heads, T, K, qh * qw).permute(0, 1, 2, 4, 3).reshape(B, heads, Q, K)
    w = _rd(kw["w"], torch.arange(heads * 2 * heads)).view(heads, 2 * heads)
    out = torch.einsum("oh,bhqk->boqk", w, torch.cat([cur, up], 1)) + _rd(kw["bias"], torch.arange(heads))[None, :, None, None]
    _wr(kw["out"], ci, out.reshape(B * heads * Q, K))


def attn_msg_bwd(**kw):
    """xargs = [dout, dcur, dup, dw (written), dbias (written), workspace]"""
    dout, dcur, dup, dw, dbias = kw["xargs"][:5]
    B, heads, T, qh, qw, K, ldk, ldkp = (kw[k] for k in ("B", "heads", "T", "qh", "qw", "K", "ldk", "ldkp"))
    Q, sh, sw = T * qh * qw, qh // 2, qw // 2
    Qp = T * sh * sw
    ci = torch.arange(B * heads * Q)[:, None] * ldk + torch.arange(K)[None, :]
    cur = _rd(kw["cur"], ci).view(B, heads, Q, K)
    g = _rd(dout, ci).view(B, heads, Q, K)
    pi = torch.arange(B * heads * Qp)[:, None] * ldkp + torch.arange(K)[None, :]
    prev = _rd(kw["prev"], pi).view(B, heads, T, sh, sw, K)
    up = torch.nn.functional.interpolate(prev.permute(0, 1, 2, 5, 3, 4).reshape(B * heads * T, K, sh, sw), scale_factor=2,
                                         mode="bilinear", align_corners=False)
    up = up.view(B, heads, T, K, qh * qw).permute(0, 1, 2, 4, 3).reshape(B, heads, Q, K)
    w = _rd(kw["w"], torch.arange(heads * 2 * heads)).view(heads, 2 * heads)
    _wr(dcur, ci, torch.einsum("oh,boqk->bhqk", w[:, :heads], g).reshape(B * heads * Q, K))
    _wr(dup, ci, torch.einsum("oh,boqk->bhqk", w[:, heads:], g).reshape(B * heads * Q, K))
    wi = torch.arange(heads * 2 * heads)
    _wr(dw, wi, torch.einsum("boqk,bhqk->oh", g, torch.cat([cur, up], 1)).reshape(-1))
    bi = torch.arange(heads)
    _wr(dbias, bi, g.sum((0, 2, 3)))


def convt3x3s2_gather(**kw):
    B, H, W, Cop = kw["B"], kw["H"], kw["W"], kw["Cop"]
    ya = _rd(kw["yall"], torch.arange(B * H * W * 9 * Cop)).view(B, H, W, 3, 3, Cop)
    out = torch.zeros(B, 2 * H, 2 * W, Cop, dtype=torch.float64) + _rd(kw["bias"], torch.arange(Cop))
    for ky in range(3):
        for kx in range(3):
            for iy in range(H):
                oy = 2 * iy - 1 + ky
                if oy < 0 or oy >= 2 * H:
                    continue
                for ix in range(W):
                    ox = 2 * ix - 1 + kx
                    if 0 <= ox < 2 * W:
                        out[:, oy, ox] += ya[:, iy, ix, ky, kx]
    _wr(kw["out"], torch.arange(out.numel()), out.reshape(-1))


def _raw_f32(addr, n):
    """fp32 view of host memory at a raw address (the optimizer entry points take arrays of tensor pointers)."""
    import ctypes
    import numpy as np
    return np.ctypeslib.as_array((ctypes.c_float * int(n)).from_address(int(addr)))


def _raw(addr, count, code):
    """typed numpy view of host memory at a raw address (code: F32 -> float32, else uint16 = bf16 bit patterns)."""
    import ctypes
    import numpy as np
    ct = ctypes.c_float if code == F32 else ctypes.c_uint16
    return np.ctypeslib.as_array((ct * int(count)).from_address(int(addr)))


def segcopy(**kw):
    """mtt_segcopy (include/mtt_hip.h): multi-segment strided copy / cast driven by the device tables; padding is not touched."""
    import numpy as np
    table = kw["table"].tolist()
    used = sorted(set(kw["chunk_seg"].tolist()))
    assert used == list(range(len(table))), "every segment must be covered by the chunk table"
    assert kw["n_chunks"] == kw["chunk_seg"].numel() == kw["chunk_off"].numel()
    for g in table:
        sa, da, la, total, n1, n2, s0, s1, s2, d0, d1, d2, sdt, ddt, vec, _ = g
        if g[15]:                                     # transposing segment: `total` counts 64 x 64 tile slots; same element mapping
            assert s1 == 1 and d2 == 1 and not vec
            n0 = total // (4096 * ((n1 + 63) // 64) * ((n2 + 63) // 64))
        else:
            n0 = total // (n1 * n2)
        i0, i1, i2 = np.meshgrid(np.arange(n0), np.arange(n1), np.arange(n2), indexing="ij")
        so = (i0 * s0 + i1 * s1 + i2 * s2).ravel()
        do = (i0 * d0 + i1 * d1 + i2 * d2).ravel()
        if vec:
            assert n2 % 4 == 0 and s0 % 4 == 0 and s1 % 4 == 0 and d0 % 4 == 0 and d1 % 4 == 0 and (sa + kw["src_base"]) % 16 == 0 \
                and (da + kw["dst_base"]) % 16 == 0, "vec segment violates its alignment contract"
        src = _raw(sa + kw["src_base"], so.max() + 1, sdt)[so]
        v = torch.from_numpy(src.astype(np.float32)) if sdt == F32 else (torch.from_numpy(src.astype(np.int32)) << 16).view(torch.float32)
        if ddt == F32:
            _raw(da + kw["dst_base"], do.max() + 1, F32)[do] = v.numpy()
        else:
            hi = v.to(torch.bfloat16)
            _raw(da + kw["dst_base"], do.max() + 1, BF16)[do] = hi.view(torch.int16).numpy().view(np.uint16)
            if ddt == SPLIT:
                lo = (v - hi.float()).to(torch.bfloat16)
                _raw(la + kw["dst_base"], do.max() + 1, BF16)[do] = lo.view(torch.int16).numpy().view(np.uint16)


def _adam_tables(kw):
    return [t.tolist() if t is not None else None for t in (kw["grads"], kw["params"], kw["exp_avg"], kw["exp_avg_sq"], kw["numel"])]


def grad_sqnorm(**kw):
    grads, _, _, _, numel = _adam_tables(kw)
    tot = 0.0
    for g, n in zip(grads, numel):
        tot += float((_raw_f32(g, n).astype("float64") ** 2).sum())
    out = kw["xargs"][0]
    _wr(out, torch.arange(1), _rd(out, torch.arange(1)) + tot)


def adam_step(**kw):
    import numpy as np
    grads, params, ms, vs, numel = _adam_tables(kw)
    total = kw["xargs"][0]
    coef = 1.0
    if kw["max_norm"] > 0 and total is not None:
        coef = min(1.0, kw["max_norm"] / (float(_rd(total, torch.arange(1))[0]) ** 0.5 + 1e-6))
    step_size, inv_sqrt_bc2 = float(np.float32(kw["step_size"])), float(np.float32(kw["inv_sqrt_bc2"]))      # f32 descriptor fields
    if kw.get("hyper") is not None:                      # device pair that replaces the by-value fields (graph-capturable step)
        step_size, inv_sqrt_bc2 = (float(v) for v in _rd(kw["hyper"], torch.arange(2)))
    for g, p, m, v, n in zip(grads, params, ms, vs, numel):
        ga, pa, ma, va = (_raw_f32(a, n) for a in (g, p, m, v))
        gg = ga.astype("float64") * coef + kw["weight_decay"] * pa.astype("float64")
        mm = kw["beta1"] * ma.astype("float64") + (1 - kw["beta1"]) * gg
        vv = kw["beta2"] * va.astype("float64") + (1 - kw["beta2"]) * gg * gg
        pa[:] = (pa.astype("float64") - step_size * mm / (np.sqrt(vv) * inv_sqrt_bc2 + kw["eps"])).astype("float32")
        ma[:] = mm.astype("float32")
        va[:] = vv.astype("float32")


def _loss_views(kw):
    B, HW, C, Cl = kw["B"], kw["HW"], kw["C"], kw["Cl"]
    x = _rd(kw["pred"], torch.arange(B * C * HW)).view(B, C, HW) if kw.get("pred") is not None else None
    y = _rd(kw["label"], torch.arange(B * Cl * HW)).view(B, Cl, HW)
    valid = (y != kw["ignore"]).all(1)                                  # [B, HW]
    return x, y, valid


def loss_label_stats(**kw):
    _, y, valid = _loss_views(kw)
    st = kw["xargs"][0]
    _wr(st, torch.arange(2), _rd(st, torch.arange(2)) + torch.stack([valid.sum().double(), y[:, 0][valid].sum()]))


def _loss_value(x, y, valid, kw, stats):
    """Task loss as a differentiable fp64 torch expression (same algebra as losses.py / the reference)."""
    kind = kw["kind"]
    n = stats[0].clamp_min(1)
    if kind <= 1:
        lab = torch.where(valid, y[:, 0], torch.zeros_like(y[:, 0])).long()
        w = None
        if kind == 1:
            wpos = (n - stats[1]) / n
            w = torch.stack([1 - wpos, wpos])
        per = torch.nn.functional.cross_entropy(x, lab, weight=w, reduction="none")
        return (per * valid).sum() / n
    if kind == 2:
        pwt = kw["pos_weight"]
        factor = 1.0 / (1.0 - pwt)
        yy = y[:, 0]
        per = torch.nn.functional.binary_cross_entropy_with_logits(x[:, 0], yy, pos_weight=torch.tensor(pwt * factor, dtype=x.dtype), reduction="none")
        return (per * valid).sum() / n / factor
    o = torch.nn.functional.normalize(x, p=2, dim=1) if kind == 4 else x
    return ((o - y).abs() * valid[:, None]).sum() / n


def loss_fwd(**kw):
    x, y, valid = _loss_views(kw)
    stats = _rd(kw["stats"], torch.arange(2))
    v = _loss_value(x, y, valid, kw, stats)
    _wr(kw["loss"], torch.arange(1), _rd(kw["loss"], torch.arange(1)) + v.reshape(1))


def loss_bwd(**kw):
    x, y, valid = _loss_views(kw)
    stats = _rd(kw["stats"], torch.arange(2))
    with torch.enable_grad():
        xr = x.clone().requires_grad_(True)
        (g,) = torch.autograd.grad(_loss_value(xr, y, valid, kw, stats), xr)
    g = g * float(_rd(kw["xargs"][0], torch.arange(1))[0])
    _wr(kw["dpred"], torch.arange(g.numel()), g.reshape(-1))


def dwconv3x3s2_bwd(**kw):
    dy, dx, dw = kw["xargs"]
    Z, B, H, W, ld = kw["Z"], kw["B"], kw["H"], kw["W"], kw["ld"]
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    x = _rd(kw["x"], torch.arange(Z * B * H * W * ld)).view(Z, B, H, W, ld).permute(0, 1, 4, 2, 3)
    w = _rd(kw["w"], torch.arange(Z * 9 * ld)).view(Z, 3, 3, ld).permute(0, 3, 1, 2)
    g = _rd(dy, torch.arange(Z * B * Ho * Wo * ld)).view(Z, B, Ho, Wo, ld).permute(0, 1, 4, 2, 3)
    dxs, dws = [], []
    for z in range(Z):
        xi = x[z].clone().requires_grad_(True); wi = w[z][:, None].clone().requires_grad_(True)
        y = torch.nn.functional.conv2d(xi, wi, None, stride=2, padding=1, groups=ld)
        with torch.enable_grad():
            pass
        gx, gw = torch.autograd.grad(y, (xi, wi), g[z])
        dxs.append(gx.permute(0, 2, 3, 1).reshape(-1)); dws.append(gw[:, 0].permute(1, 2, 0).reshape(-1))
    if dx is not None:
        v = torch.cat(dxs); _wr(dx, torch.arange(v.numel()), v)
    if dw is not None:
        v = torch.cat(dws); _wr(dw, torch.arange(v.numel()), v)


def avgpool_ceil_bwd(**kw):
    dy, dx = kw["xargs"]
    B, H, W, k, ld = kw["B"], kw["H"], kw["W"], kw["k"], kw["ld"]
    Ho, Wo = -(-H // k), -(-W // k)
    xi = torch.zeros(B, ld, H, W, dtype=torch.float64, requires_grad=True)
    y = torch.nn.functional.avg_pool2d(xi, k, k, 0, ceil_mode=True)
    g = _rd(dy, torch.arange(B * Ho * Wo * ld)).view(B, Ho, Wo, ld).permute(0, 3, 1, 2)
    (gx,) = torch.autograd.grad(y, xi, g)
    _wr(dx, torch.arange(B * H * W * ld), gx.permute(0, 2, 3, 1).reshape(-1))


def convt3x3s2_gather_bwd(**kw):
    dout, dyall = kw["xargs"]
    B, H, W, Cop = kw["B"], kw["H"], kw["W"], kw["Cop"]
    g = _rd(dout, torch.arange(B * 4 * H * W * Cop)).view(B, 2 * H, 2 * W, Cop)
    out = torch.zeros(B, H, W, 3, 3, Cop, dtype=torch.float64)
    for ky in range(3):
        for kx in range(3):
            for iy in range(H):
                oy = 2 * iy - 1 + ky
                if not 0 <= oy < 2 * H:
                    continue
                for ix in range(W):
                    ox = 2 * ix - 1 + kx
                    if 0 <= ox < 2 * W:
                        out[:, iy, ix, ky, kx] = g[:, oy, ox]
    _wr(dyall, torch.arange(out.numel()), out.reshape(-1))


def _upconv_expand_math(z, h, w):
    """z [n, h, w, 3, 3, Cp] (tap planes W[ky,kx] x) -> [n, 4h, 4w, Cp]: sum over taps of the x4 bilinear expansion of plane (ky, kx)
    read at (Y + ky - 1, X + kx - 1), zero outside the 4h x 4w map — mtt_upconv_desc, restated with F.interpolate + zero-padded
    shifts (independent of the kernels' separable constant-weight form)."""
    n, Cp = z.shape[0], z.shape[-1]
    out = torch.zeros(n, Cp, 4 * h, 4 * w, dtype=z.dtype)
    for ky in range(3):
        for kx in range(3):
            up = torch.nn.functional.interpolate(z[:, :, :, ky, kx].permute(0, 3, 1, 2), scale_factor=4, mode="bilinear", align_corners=False)
            out = out + torch.nn.functional.pad(up, (1, 1, 1, 1))[:, :, ky:ky + 4 * h, kx:kx + 4 * w]
    return out.permute(0, 2, 3, 1)


def upconv4_expand(**kw):
    Z, B, h, w, Cn, Cp = (kw[k] for k in ("Z", "B", "h", "w", "C", "Cp"))
    n = Z * B
    z = _rd(kw["z"], torch.arange(n * h * w * 9 * Cp)).view(n, h, w, 3, 3, Cp)
    y = _upconv_expand_math(z, h, w).reshape(Z, B * 16 * h * w, Cp)
    sc = _rd(kw["colscale"], torch.arange(Z * Cn)).view(Z, 1, Cn) if kw.get("colscale") is not None else 1.0
    bs = _rd(kw["bias"], torch.arange(Z * Cn)).view(Z, 1, Cn) if kw.get("bias") is not None else 0.0
    out = torch.zeros_like(y)
    out[..., :Cn] = _act(y[..., :Cn] * sc + bs, kw.get("act", 0) or 0)
    _wr(kw["y"], torch.arange(out.numel()), out.reshape(-1))


def upconv4_gather(**kw):
    Z, B, h, w, Cp = (kw[k] for k in ("Z", "B", "h", "w", "Cp"))
    n = Z * B
    g = _rd(kw["y"], torch.arange(n * 16 * h * w * Cp)).view(n, 4 * h, 4 * w, Cp)
    z = torch.zeros(n, h, w, 3, 3, Cp, dtype=torch.float64, requires_grad=True)
    (dz,) = torch.autograd.grad(_upconv_expand_math(z, h, w), z, g)
    _wr(kw["z"], torch.arange(dz.numel()), dz.reshape(-1))


# ---- TaskPrompter-Swin forward entry points ---------------------------------------------------------------------------------
def patchify(args):
    img, cols, B, H, W, P, ldc, odt = args[:8]
    gh, gw = H // P, W // P
    x = img.double().reshape(B, 3, gh, P, gw, P).permute(0, 2, 4, 1, 3, 5).reshape(B * gh * gw, 3 * P * P)
    out = torch.zeros(B * gh * gw, ldc, dtype=torch.float64)
    out[:, :3 * P * P] = x
    _wr(cols, torch.arange(out.numel()), out.reshape(-1))


def resize_nchw(args):
    src, dst, planes, Hin, Win, Hout, Wout = args[:7]
    x = _rd(src, torch.arange(planes * Hin * Win)).view(1, planes, Hin, Win)
    y = torch.nn.functional.interpolate(x, (Hout, Wout), mode="bilinear", align_corners=False)
    _wr(dst, torch.arange(planes * Hout * Wout), y.reshape(-1))


def gather_rows(**kw):
    B, rows, Cn = kw["B"], kw["rows"], kw["C"]
    idx = kw["idx"]
    fi, oi = flat(idx)
    b = torch.arange(B)[:, None, None]
    r = torch.arange(rows)[None, :, None]
    c = torch.arange(Cn)[None, None, :]
    src_row = fi[oi + b * (kw.get("idx_bs") or 0) + r].long()                      # [B, rows, 1]
    ok = (src_row >= 0).expand(B, rows, Cn)
    v = _rd(kw["src"], b * (kw.get("src_bs") or 0) + src_row * kw["ld_src"] + c, ok)
    di = (b * (kw.get("dst_bs") or 0) + r * kw["ld_dst"] + c).expand(B, rows, Cn)
    if kw.get("skip_neg"):
        _wr(kw["dst"], di[ok], v[ok])
    else:
        _wr(kw["dst"], di.reshape(-1), v.reshape(-1))


def winattn_fwd(**kw):
    nwin, nW, nH, T, ws2 = (kw[k] for k in ("nwin", "nW", "nH", "T", "ws2"))
    N, Cn = T + ws2, nH * 32
    qkv = _rd(kw["qkv"], torch.arange(nwin * N * 3 * Cn)).view(nwin, N, 3, nH, 32)
    if kw.get("dtype", F32) == BF16:
        qkv = _bf16_round(qkv)
    q, k, v = (qkv[:, :, i].permute(0, 2, 1, 3) for i in range(3))                # [nwin, nH, N, 32]
    raw = q @ k.transpose(-1, -2)
    S = raw * kw["scale"]
    bias = _rd(kw["bias"], torch.arange(nH * ws2 * ws2)).view(1, nH, ws2, ws2)
    S[:, :, T:, T:] += bias
    if kw.get("mask") is not None:
        mask = _rd(kw["mask"], torch.arange(nW * ws2 * ws2)).view(nW, 1, ws2, ws2)
        S[:, :, T:, T:] += mask.repeat(nwin // nW, 1, 1, 1)
    o = (torch.softmax(S, -1) @ v).permute(0, 2, 1, 3).reshape(nwin * N * Cn)
    _wr(kw["out"], torch.arange(nwin * N * Cn), o)
    if kw.get("rawmap") is not None:
        fp, op = flat(kw["pix"])
        pix = fp[op:op + nW * ws2].long().view(nW, ws2)
        w = torch.arange(nwin)
        px = pix[w % nW]                                                          # [nwin, ws2]
        b = (w // nW)[:, None, None, None]
        h = torch.arange(nH)[None, :, None, None]
        t = torch.arange(T)[None, None, :, None]
        dst = ((b * nH + h) * T + t) * kw["map_ld"] + kw.get("map_off", 0) + px[:, None, None, :]
        ok = (px >= 0)[:, None, None, :].expand(nwin, nH, T, ws2)
        _wr(kw["rawmap"], dst[ok], raw[:, :, :T, T:][ok])


def winattn_bwd(**kw):
    """autograd of the forward contract: dqkv, and dS of the window x window part (= gradient wrt a per-window additive term)."""
    dout, drawmap, dqkv, dS_out = kw["xargs"]
    nwin, nW, nH, T, ws2 = (kw[k] for k in ("nwin", "nW", "nH", "T", "ws2"))
    N, Cn = T + ws2, nH * 32
    qkv = _rd(kw["qkv"], torch.arange(nwin * N * 3 * Cn)).view(nwin, N, 3, nH, 32).clone().requires_grad_(True)
    extra = torch.zeros(nwin, nH, ws2, ws2, dtype=torch.float64, requires_grad=True)
    const = _rd(kw["bias"], torch.arange(nH * ws2 * ws2)).view(1, nH, ws2, ws2).expand(nwin, nH, ws2, ws2)
    if kw.get("mask") is not None:
        const = const + _rd(kw["mask"], torch.arange(nW * ws2 * ws2)).view(nW, 1, ws2, ws2).repeat(nwin // nW, 1, 1, 1)
    q, k, v = (qkv[:, :, i].permute(0, 2, 1, 3) for i in range(3))
    raw = q @ k.transpose(-1, -2)
    S = raw * kw["scale"] + torch.nn.functional.pad(const + extra, (T, 0, T, 0))
    o = (torch.softmax(S, -1) @ v).permute(0, 2, 1, 3)                                  # [nwin, N, nH, 32]
    loss = (o * _rd(dout, torch.arange(nwin * N * Cn)).view(nwin, N, nH, 32)).sum()
    if drawmap is not None:
        fp, op = flat(kw["pix"])
        pix = fp[op:op + nW * ws2].long().view(nW, ws2)
        w = torch.arange(nwin)
        px = pix[w % nW]
        b = (w // nW)[:, None, None, None]
        h = torch.arange(nH)[None, :, None, None]
        t = torch.arange(T)[None, None, :, None]
        src = ((b * nH + h) * T + t) * kw["map_ld"] + kw.get("map_off", 0) + px[:, None, None, :]
        ok = (px >= 0)[:, None, None, :].expand(nwin, nH, T, ws2)
        loss = loss + (raw[:, :, :T, T:] * _rd(drawmap, src, ok)).sum()
    gq, gs = torch.autograd.grad(loss, [qkv, extra])
    _wr(dqkv, torch.arange(nwin * N * 3 * Cn), gq.reshape(-1))
    if dS_out is not None:
        _wr(dS_out, torch.arange(nwin * nH * ws2 * ws2), gs.reshape(-1))


def chanattn_fwd(**kw):
    B, T, Cn, ce, nh, nw = (kw[k] for k in ("B", "T", "C", "ce", "nh", "nw"))
    ldk = kw["ldk"]
    q = _rd(kw["q"], torch.arange(B * T * ce)).view(B, T, ce)
    rows = torch.arange(B * 2 * ce)[:, None] * ldk + torch.arange(Cn)[None, :]
    kv = _rd(kw["kvT"], rows).view(B, 2, ce, Cn)
    if kw.get("kvbias") is not None:
        kv = kv + _rd(kw["kvbias"], torch.arange(2 * ce)).view(1, 2, ce, 1)
    kT, vT = kv[:, 0], kv[:, 1]
    r = math.isqrt(ce)
    wh, ww = r // nh, r // nw

    def split(t):                                                                  # [B, X, ce] -> [B, nwin, X, wh*ww]
        return t.view(B, t.shape[1], nh, wh, nw, ww).permute(0, 2, 4, 1, 3, 5).reshape(B, nh * nw, t.shape[1], wh * ww)
    q_, k_, v_ = split(q), split(kT.transpose(1, 2)), split(vT.transpose(1, 2))    # k_: [B, nwin, C, P]
    raw = q_ @ k_.transpose(-1, -2)                                                # [B, nwin, T, C]
    cx = torch.softmax(raw * kw["scale"], -1) @ v_                                 # [B, nwin, T, P]
    cx = cx.view(B, nh, nw, T, wh, ww).permute(0, 3, 1, 4, 2, 5).reshape(B * T * ce)
    _wr(kw["rawchan"], torch.arange(B * T * nh * nw * Cn), raw.permute(0, 2, 1, 3).reshape(-1))
    _wr(kw["cx"], torch.arange(B * T * ce), cx)


def conv3s2_nchw(**kw):
    B, Ci, Co, H, W = (kw[k] for k in ("B", "Ci", "Co", "H", "W"))
    g = lambda k: kw.get(k) or 0
    b = torch.arange(B)[:, None, None]
    c = torch.arange(Ci)[None, :, None]
    pxl = torch.arange(H * W)[None, None, :]
    x = _rd(kw["x"], b * g("x_bs") + c * g("x_cs") + g("x_off") + pxl).view(B, Ci, H, W)
    w = _rd(kw["w"], torch.arange(Co * Ci * 9)).view(Co, Ci, 3, 3)
    bias = _rd(kw["bias"], torch.arange(Co)) if kw.get("bias") is not None else None
    y = torch.nn.functional.conv2d(x, w, bias, stride=2, padding=1)
    co = torch.arange(Co)[None, :, None]
    po = torch.arange((H // 2) * (W // 2))[None, None, :]
    _wr(kw["y"], (b * g("y_bs") + co * g("y_cs") + g("y_off") + po).reshape(-1), y.reshape(-1))


def chanattn_bwd(**kw):
    """xargs = [drawchan or None, dcx, dq (written), dkvT (written, pitch ldg), ldg, workspace]: autograd of the emulated forward."""
    drawchan, dcx, dq, dkvT, ldg = kw["xargs"][:5]
    B, T, Cn, ce, nh, nw = (kw[k] for k in ("B", "T", "C", "ce", "nh", "nw"))
    ldk = kw["ldk"]
    q = _rd(kw["q"], torch.arange(B * T * ce)).view(B, T, ce).requires_grad_(True)
    rows = torch.arange(B * 2 * ce)[:, None] * ldk + torch.arange(Cn)[None, :]
    kv = _rd(kw["kvT"], rows).view(B, 2, ce, Cn).requires_grad_(True)
    kT, vT = kv[:, 0], kv[:, 1]
    r = math.isqrt(ce)
    wh, ww = r // nh, r // nw

    def split(t):
        return t.view(B, t.shape[1], nh, wh, nw, ww).permute(0, 2, 4, 1, 3, 5).reshape(B, nh * nw, t.shape[1], wh * ww)
    q_, k_, v_ = split(q), split(kT.transpose(1, 2)), split(vT.transpose(1, 2))
    raw = q_ @ k_.transpose(-1, -2)
    cx = torch.softmax(raw * kw["scale"], -1) @ v_
    cx = cx.view(B, nh, nw, T, wh, ww).permute(0, 3, 1, 4, 2, 5).reshape(B, T, ce)
    outs, grads = [cx], [_rd(dcx, torch.arange(B * T * ce)).view(B, T, ce)]
    if drawchan is not None:
        outs.append(raw.permute(0, 2, 1, 3))
        grads.append(_rd(drawchan, torch.arange(B * T * nh * nw * Cn)).view(B, T, nh * nw, Cn))
    gq, gkv = torch.autograd.grad(outs, [q, kv], grads)
    _wr(dq, torch.arange(B * T * ce), gq.reshape(-1))
    _wr(dkvT, (torch.arange(B * 2 * ce)[:, None] * ldg + torch.arange(Cn)[None, :]).reshape(-1), gkv.reshape(-1))


def conv3s2_nchw_bwd(**kw):
    """xargs = [dy (laid out as y), dx (as x; or None), dw [Co, Ci, 3, 3] or None, db [Co] or None]"""
    dy, dx, dw, db = kw["xargs"][:4]
    B, Ci, Co, H, W = (kw[k] for k in ("B", "Ci", "Co", "H", "W"))
    g = lambda k: kw.get(k) or 0
    b = torch.arange(B)[:, None, None]
    c = torch.arange(Ci)[None, :, None]
    pxl = torch.arange(H * W)[None, None, :]
    xi = b * g("x_bs") + c * g("x_cs") + g("x_off") + pxl
    x = _rd(kw["x"], xi).view(B, Ci, H, W).requires_grad_(True)
    w = _rd(kw["w"], torch.arange(Co * Ci * 9)).view(Co, Ci, 3, 3).requires_grad_(True)
    bias = torch.zeros(Co, dtype=torch.float64, requires_grad=True)
    y = torch.nn.functional.conv2d(x, w, bias, stride=2, padding=1)
    co = torch.arange(Co)[None, :, None]
    po = torch.arange((H // 2) * (W // 2))[None, None, :]
    gy = _rd(dy, b * g("y_bs") + co * g("y_cs") + g("y_off") + po).view(B, Co, H // 2, W // 2)
    gx, gw, gb = torch.autograd.grad(y, [x, w, bias], gy)
    if dx is not None:
        _wr(dx, xi.reshape(-1), gx.reshape(-1))
    if dw is not None:
        _wr(dw, torch.arange(Co * Ci * 9), gw.reshape(-1))
        if db is not None:
            _wr(db, torch.arange(Co), gb)


def boxes_overlap_bev(args):
    from . import iou3d_oracle
    a, na, b, nb, out, iou = args[:6]
    r = iou3d_oracle.pairwise(a.reshape(na, 5).numpy(), b.reshape(nb, 5).numpy(), bool(iou))
    _wr(out, torch.arange(na * nb), torch.from_numpy(r).reshape(-1))


def nms_bev(args):
    from . import iou3d_oracle
    boxes, n, thresh, rotated, keep, num_out, ws = args[:7]
    k = iou3d_oracle.nms(boxes.reshape(n, 5).numpy(), thresh, bool(rotated))
    _wr(keep, torch.arange(len(k)), torch.from_numpy(k))
    _wr(num_out, torch.arange(1), torch.tensor([len(k)]))


_TABLE = dict(gather_rows=gather_rows, winattn_fwd=winattn_fwd, winattn_bwd=winattn_bwd, chanattn_fwd=chanattn_fwd, conv3s2_nchw=conv3s2_nchw, gemm=gemm, upconv4_expand=upconv4_expand, upconv4_gather=upconv4_gather, attn_fwd=attn_fwd, softmax_fwd=softmax_fwd, softmax_bwd=softmax_bwd,
              layernorm_fwd=layernorm_fwd, layernorm_bwd=layernorm_bwd, chan_logits=chan_logits, modulate=modulate,
              ctr_mix=ctr_mix, bilinear_fwd=bilinear_fwd, bilinear_bwd=bilinear_bwd, bn_stats=bn_stats,
              bn_apply=bn_apply, bn_bwd_reduce=bn_bwd_reduce, bn_bwd_apply=bn_bwd_apply,
              modulate_bwd=modulate_bwd, chan_logits_bwd=chan_logits_bwd, ctr_dw=ctr_dw,
              dwconv3x3s2=dwconv3x3s2, avgpool_ceil=avgpool_ceil, layernorm_mt=layernorm_mt, attn_msg=attn_msg, attn_msg_bwd=attn_msg_bwd,
              convt3x3s2_gather=convt3x3s2_gather, dwconv3x3s2_bwd=dwconv3x3s2_bwd, avgpool_ceil_bwd=avgpool_ceil_bwd,
              convt3x3s2_gather_bwd=convt3x3s2_gather_bwd, attn_bwd=attn_bwd, grad_sqnorm=grad_sqnorm, adam_step=adam_step, loss_label_stats=loss_label_stats,
              loss_fwd=loss_fwd, loss_bwd=loss_bwd, chanattn_bwd=chanattn_bwd, conv3s2_nchw_bwd=conv3s2_nchw_bwd, segcopy=segcopy,
              ctr_weights=ctr_weights, ctr_weights_bwd=ctr_weights_bwd, detloss_fwd=detloss_fwd, detloss_bwd=detloss_bwd)
_POS = dict(boxes_overlap_bev=boxes_overlap_bev, nms_bev=nms_bev, patchify=patchify, resize_nchw=resize_nchw, patchify16=patchify16, cast2d=cast2d, split_cast=split_cast, pixshuf2=pixshuf2, colsum=colsum, colsum_batched=colsum_batched, add_rows=add_rows, rowscale_cast=rowscale_cast, rowscale_cast_colsum=rowscale_cast_colsum)


def call(name, **kw):
    with torch.enable_grad() if name in ("dwconv3x3s2_bwd", "avgpool_ceil_bwd", "loss_bwd", "upconv4_gather", "winattn_bwd", "chanattn_bwd", "conv3s2_nchw_bwd", "ctr_weights_bwd", "detloss_bwd") else torch.no_grad():
        if name in _POS:
            return _POS[name](kw["args"])
        return _TABLE[name](**kw)
