"""CPU: the C-ABI emulator (the per-entry-point oracle) against torch.nn.functional — this is what makes
the emulator independent of the kernels it later judges."""
import torch
import torch.nn.functional as F

from oracle import abi_emul as E

E_F32 = 0


def g_(seed=0):
    return torch.Generator().manual_seed(seed)


def test_gemm_linear_and_epilogue():
    g = g_()
    x, w, b = torch.randn(50, 40, generator=g), torch.randn(30, 40, generator=g), torch.randn(30, generator=g)
    res = torch.randn(50, 32, generator=g)
    out = torch.zeros(50, 32)
    E.call("gemm", A=x, B=w, D=out, M=50, N=30, K=40, a_op=0, b_op=0, prec=1, lda=40, ldb=40, ldd=32, alpha=1.0,
           colshift=b, act=1, resid=res, ldr=32, n_store=32)
    ref = F.gelu(F.linear(x, w, b)) + res[:, :30]
    assert torch.allclose(out[:, :30], ref, atol=1e-5) and float(out[:, 30:].abs().max()) == 0.0


def test_gemm_dgrad_wgrad_layouts():
    g = g_(1)
    x, w, dy = torch.randn(50, 40, generator=g), torch.randn(24, 40, generator=g), torch.randn(50, 24, generator=g)
    dx, dw = torch.zeros(50, 40), torch.zeros(24, 40)
    E.call("gemm", A=dy, B=w, D=dx, M=50, N=40, K=24, a_op=0, b_op=1, prec=1, lda=24, ldb=40, ldd=40, alpha=1.0)
    E.call("gemm", A=dy, B=x, D=dw, M=24, N=40, K=50, a_op=1, b_op=1, prec=1, lda=24, ldb=40, ldd=40, alpha=1.0)
    assert torch.allclose(dx, dy @ w, atol=1e-5) and torch.allclose(dw, dy.t() @ x, atol=1e-5)


def _pack(w):   # [Co,Ci,3,3] -> [Co, 9*Cp]
    Co, Ci = w.shape[:2]
    Cp = (Ci + 7) // 8 * 8
    buf = torch.zeros(Co, 9, Cp)
    buf[:, :, :Ci] = w.permute(0, 2, 3, 1).reshape(Co, 9, Ci)
    return buf.reshape(Co, 9 * Cp), Cp


def test_gemm_conv3x3_fwd_dgrad_wgrad():
    g = g_(2)
    B, H, W, Ci, Co = 2, 6, 5, 12, 10
    for dil in (1, 2):
        x = torch.randn(B, Ci, H, W, generator=g, requires_grad=True)
        w = torch.randn(Co, Ci, 3, 3, generator=g, requires_grad=True)
        y = F.conv2d(x, w, padding=dil, dilation=dil)
        dy = torch.randn(y.shape, generator=g)
        y.backward(dy)
        wp, Cp = _pack(w.detach())
        xn = torch.zeros(B * H * W, Cp); xn[:, :Ci] = x.detach().permute(0, 2, 3, 1).reshape(-1, Ci)
        out = torch.zeros(B * H * W, 16)
        geom = dict(H=H, W=W, C=Ci, Cp=Cp, dil=dil, flip=0)
        E.call("gemm", A=xn, B=wp, D=out, M=B * H * W, N=Co, K=9 * Cp, a_op=2, b_op=0, prec=1, lda=Cp, ldb=9 * Cp, ldd=16, alpha=1.0, conv=geom)
        assert torch.allclose(out[:, :Co], y.detach().permute(0, 2, 3, 1).reshape(-1, Co), atol=1e-4)
        # dgrad: weights packed [Ci, 9*Cop] (k = tap*Cop + co), taps mirrored by the loader
        Cop = 16
        wd = torch.zeros(Ci, 9, Cop); wd[:, :, :Co] = w.detach().permute(1, 2, 3, 0).reshape(Ci, 9, Co)
        dyn = torch.zeros(B * H * W, Cop); dyn[:, :Co] = dy.permute(0, 2, 3, 1).reshape(-1, Co)
        dxo = torch.zeros(B * H * W, Cp)
        E.call("gemm", A=dyn, B=wd.reshape(Ci, 9 * Cop), D=dxo, M=B * H * W, N=Ci, K=9 * Cop, a_op=2, b_op=0, prec=1, lda=Cop,
               ldb=9 * Cop, ldd=Cp, alpha=1.0, conv=dict(H=H, W=W, C=Co, Cp=Cop, dil=dil, flip=1))
        assert torch.allclose(dxo[:, :Ci], x.grad.permute(0, 2, 3, 1).reshape(-1, Ci), atol=1e-4)
        dwo = torch.zeros(Co, 9 * Cp)
        E.call("gemm", A=dyn, B=xn, D=dwo, M=Co, N=9 * Cp, K=B * H * W, a_op=1, b_op=3, prec=1, lda=Cop, ldb=Cp, ldd=9 * Cp,
               alpha=1.0, conv=geom)
        ref = w.grad.permute(0, 2, 3, 1).reshape(Co, 9, Ci)
        assert torch.allclose(dwo.reshape(Co, 9, Cp)[:, :, :Ci], ref, atol=1e-3)


def test_gemm_pixshuf_is_conv_transpose():
    g = g_(3)
    B, H, W, Ci, Co = 2, 3, 4, 8, 5
    x, w, b = torch.randn(B, Ci, H, W, generator=g), torch.randn(Ci, Co, 2, 2, generator=g), torch.randn(Co, generator=g)
    ref = F.conv_transpose2d(x, w, b, stride=2).permute(0, 2, 3, 1).reshape(-1, Co)
    out = torch.zeros(B * 4 * H * W, 8)
    E.call("gemm", A=x.permute(0, 2, 3, 1).reshape(-1, Ci).contiguous(), B=w.permute(2, 3, 1, 0).reshape(4 * Co, Ci).contiguous(), D=out,
           M=B * H * W, N=4 * Co, K=Ci, a_op=0, b_op=0, prec=1, lda=Ci, ldb=Ci, ldd=8, alpha=1.0, colshift=b.repeat(4),
           store_mode=1, ps_H=H, ps_W=W, ps_Co=Co)
    assert torch.allclose(out[:, :Co], ref, atol=1e-5)


def test_attention_and_bilinear_and_ln():
    g = g_(4)
    B, N, nH, T = 2, 20, 2, 3
    qkv = torch.randn(B * N, 3 * 128, generator=g)
    out, raw = torch.zeros(B * N, 128), torch.zeros(B, nH, T, N)
    E.call("attn_fwd", qkv=qkv, out=out, rawlog=raw, B=B, N=N, nH=nH, T=T, dtype=0, prec=1, scale=0.125)
    q, k, v = (qkv.view(B, N, 3, nH, 64)[:, :, i].transpose(1, 2) for i in range(3))
    ref = F.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(B * N, 128)
    assert torch.allclose(out, ref, atol=1e-5) and torch.allclose(raw, (q @ k.transpose(-1, -2))[:, :, :T], atol=1e-4)
    for (Hi, Wi, Ho, Wo) in ((4, 6, 16, 24), (8, 8, 4, 4), (5, 7, 11, 9)):
        x = torch.randn(2, 3, Hi, Wi, generator=g, requires_grad=True)
        ref = F.interpolate(x, size=(Ho, Wo), mode="bilinear", align_corners=False)
        xin = torch.zeros(2 * Hi * Wi, 8); xin[:, :3] = x.detach().permute(0, 2, 3, 1).reshape(-1, 3)
        o = torch.zeros(2, 3, Ho, Wo)
        E.call("bilinear_fwd", **{"in": xin}, out=o, B=2, C=3, Hin=Hi, Win=Wi, Hout=Ho, Wout=Wo, ld_in=8, ld_out=0, out_nchw=1)
        assert torch.allclose(o, ref.detach(), atol=1e-5)
        dy = torch.randn(ref.shape, generator=g)
        ref.backward(dy)
        din = torch.zeros(2 * Hi * Wi, 8)
        E.call("bilinear_bwd", **{"in": dy}, out=din, B=2, C=3, Hin=Hi, Win=Wi, Hout=Ho, Wout=Wo, ld_in=8, ld_out=0, out_nchw=1)
        assert torch.allclose(din[:, :3], x.grad.permute(0, 2, 3, 1).reshape(-1, 3), atol=1e-5)
    x = torch.randn(9, 32, generator=g, requires_grad=True)
    gam, bet = torch.randn(32, generator=g, requires_grad=True), torch.randn(32, generator=g, requires_grad=True)
    ref = F.layer_norm(x, (32,), gam, bet, 1e-6)
    y, mean, rstd = torch.zeros(9, 32), torch.zeros(9), torch.zeros(9)
    E.call("layernorm_fwd", x=x.detach(), y=y, gamma=gam.detach(), beta=bet.detach(), mean=mean, rstd=rstd, rows=9, C=32, ldx=32, ldy=32, eps=1e-6)
    assert torch.allclose(y, ref.detach(), atol=1e-5)
    dy = torch.randn(9, 32, generator=g)
    ref.backward(dy)
    dx, dg, db = torch.zeros(9, 32), torch.zeros(32), torch.zeros(32)
    E.call("layernorm_bwd", x=x.detach(), dy=dy, gamma=gam.detach(), mean=mean, rstd=rstd, dx=dx, dgamma=dg, dbeta=db, rows=9, C=32, ldx=32, ldy=32, eps=1e-6)
    assert torch.allclose(dx, x.grad, atol=1e-5) and torch.allclose(dg, gam.grad, atol=1e-5) and torch.allclose(db, bet.grad, atol=1e-5)


def test_bn_train_fwd_bwd():
    g = g_(5)
    rows, C = 40, 12
    x = torch.randn(rows, C, generator=g, requires_grad=True)
    gam, bet = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g)
    xi = x.t().reshape(1, C, rows, 1).transpose(0, 2).reshape(rows, C, 1, 1)
    ref = F.gelu(F.batch_norm(xi, None, None, gam, bet, True, 0.1, 1e-5))
    dy = torch.randn(rows, C, generator=g)
    ref.backward(dy.reshape(rows, C, 1, 1))
    xp = torch.zeros(rows, 16); xp[:, :C] = x.detach()
    mean, m2 = torch.zeros(C), torch.zeros(C)
    E.call("bn_stats", x=xp, mean_out=mean, m2_out=m2, rows=rows, C=C, ld=16, xargs=[torch.zeros(1)])
    var = m2 / rows; rstd = torch.rsqrt(var + 1e-5)
    y = torch.zeros(rows, 16)
    E.call("bn_apply", x=xp, y=y, mean=mean, rstd=rstd, gamma=gam, beta=bet, rows=rows, C=C, ld=16, act=1)
    assert torch.allclose(y[:, :C], ref.detach().reshape(rows, C), atol=1e-5)
    dyp = torch.zeros(rows, 16); dyp[:, :C] = dy
    ds, dsx = torch.zeros(C), torch.zeros(C)
    E.call("bn_bwd_reduce", x=xp, dy=dyp, mean=mean, rstd=rstd, gamma=gam, beta=bet, dsum=ds, dsumxh=dsx, rows=rows, C=C, ld=16, act=1, xargs=[torch.zeros(1)])
    dx = torch.zeros(rows, 16)
    E.call("bn_bwd_apply", x=xp, dy=dyp, dx=dx, mean=mean, rstd=rstd, gamma=gam, beta=bet, dsum=ds, dsumxh=dsx, rows=rows, C=C, ld=16, act=1)
    assert torch.allclose(dx[:, :C], x.grad, atol=1e-4)


def test_upconv_taps_first_is_upsample_then_conv():
    """mtt_upconv4_expand / _gather: (a) GEMM with the nine stacked tap matrices + expansion == F.conv2d(F.interpolate(x, 4)) and its
    autograd adjoint; (b) the kernels' algorithm — separable passes with the constant x4 phase weights, replicated-border slots and
    hi-res zero-padding masks (csrc/upconv.hip), restated here lane by lane — equals the emulator on degenerate and ragged maps."""
    torch.manual_seed(0)
    B, h, w, C = 2, 3, 5, 11
    x = torch.randn(B, C, h, w, dtype=torch.float64)
    wt = torch.randn(C, C, 3, 3, dtype=torch.float64) * 0.2
    z = torch.einsum("bchw,ocyx->bhwyxo", x, wt)                       # tap planes W[ky,kx] x
    ref = F.conv2d(F.interpolate(x, scale_factor=4, mode="bilinear"), wt, padding=1).permute(0, 2, 3, 1)
    assert (E._upconv_expand_math(z, h, w) - ref).abs().max() < 1e-12

    SA, SB = [0, 0, 0, 1, 1, 1], [1, 1, 1, 2, 2, 2]
    WA = [.625, .375, .125, .875, .625, .375]
    base = [.125, .375, .625, .875, .875, .625, .375, .125]

    def border_weights(i, n):
        wv = list(base)
        if i == 0:
            wv[0] = wv[1] = 0.0; wv[2] += .375; wv[3] += .125
        if i == n - 1:
            wv[6] = wv[7] = 0.0; wv[4] += .125; wv[5] += .375
        return wv

    for (h, w) in ((1, 1), (1, 4), (3, 2), (4, 5), (2, 1)):
        n, Cp = 2, 8
        z = torch.randn(n, h, w, 3, 3, Cp, dtype=torch.float64)
        out = torch.zeros(n, 4 * h, 4 * w, Cp, dtype=torch.float64)
        for i in range(h):
            for j in range(w):
                cs, rs = [max(j - 1, 0), j, min(j + 1, w - 1)], [max(i - 1, 0), i, min(i + 1, h - 1)]
                mx, my = {0: float(j > 0), 5: float(j < w - 1)}, {0: float(i > 0), 5: float(i < h - 1)}
                acc = torch.zeros(n, 4, 4, Cp, dtype=torch.float64)
                for dy in range(3):
                    for s in range(3):
                        v = torch.zeros(n, 4, Cp, dtype=torch.float64)
                        for dx in range(3):
                            f = [z[:, rs[s], cs[c], dy, dx] for c in range(3)]
                            for q in range(4):
                                r1 = q + dx
                                v[:, q] += mx.get(r1, 1.0) * (WA[r1] * f[SA[r1]] + (1 - WA[r1]) * f[SB[r1]])
                        for p in range(4):
                            r1 = p + dy
                            R = (WA[r1] if SA[r1] == s else ((1 - WA[r1]) if SB[r1] == s else 0.0)) * my.get(r1, 1.0)
                            acc[:, p] += R * v
                out[:, 4 * i:4 * i + 4, 4 * j:4 * j + 4] = acc
        assert (out - E._upconv_expand_math(z, h, w)).abs().max() < 1e-12

        g = torch.randn(n, 4 * h, 4 * w, Cp, dtype=torch.float64)
        zz = z.clone().requires_grad_(True)
        (dref,) = torch.autograd.grad(E._upconv_expand_math(zz, h, w), zz, g)
        dz = torch.zeros_like(z)
        for i in range(h):
            wy = border_weights(i, h)
            for j in range(w):
                wx = border_weights(j, w)
                acc = torch.zeros(n, 3, 3, Cp, dtype=torch.float64)
                for tr in range(10):
                    rho = 4 * i - 3 + tr
                    if not 0 <= rho < 4 * h:
                        continue
                    s = torch.zeros(n, 3, Cp, dtype=torch.float64)
                    for u in range(10):
                        X = 4 * j - 3 + u
                        if 0 <= X < 4 * w:
                            for dx in range(3):
                                if 0 <= u + dx - 2 < 8:
                                    s[:, dx] += wx[u + dx - 2] * g[:, rho, X]
                    for dy in range(3):
                        if 0 <= tr + dy - 2 < 8:
                            acc[:, dy] += wy[tr + dy - 2] * s
                dz[:, i, j] = acc
        assert (dz - dref).abs().max() < 1e-12


def test_bias_gradients_from_the_producers():
    """mtt_gemm_desc.colsum_out (column sums of the stored tile: the fc2 input gradient with its GELU' epilogue is fc1's output gradient)
    and mtt_rowscale_cast_colsum (the cast of the residual gradient yields the proj / fc2 bias gradients) against autograd."""
    g = g_(7)
    M, C, Hd = 40, 24, 48
    x = torch.randn(M, C, generator=g)
    W1 = (torch.randn(Hd, C, generator=g) * 0.3).requires_grad_(True)
    b1 = torch.randn(Hd, generator=g).requires_grad_(True)
    W2 = (torch.randn(C, Hd, generator=g) * 0.3).requires_grad_(True)
    b2 = torch.randn(C, generator=g).requires_grad_(True)
    z = F.linear(x, W1, b1)
    y = F.linear(F.gelu(z), W2, b2)
    dy = torch.randn(M, C, generator=g)
    y.backward(dy)
    # fc2 input gradient * GELU'(z) with the column sums taken by the "epilogue": = db1
    dz, db1 = torch.zeros(M, Hd), torch.full((Hd,), 9.0)
    E.call("gemm", A=dy, B=W2.detach(), D=dz, M=M, N=Hd, K=C, a_op=0, b_op=1, prec=1, lda=C, ldb=Hd, ldd=Hd, alpha=1.0, act=3,
           aux_in=z.detach(), aux_dtype=E_F32, ldaux=Hd, batch=1, colsum_out=db1, colsum_ws=torch.zeros(4 * Hd))
    assert torch.allclose(db1, b1.grad, atol=1e-4)
    assert torch.allclose(dz.t() @ x, W1.grad, atol=1e-4)
    # the cast of dy (here with per-sample scales of 1) yields db2
    out, db2 = torch.zeros(M, C), torch.full((C,), 9.0)
    E.call("rowscale_cast_colsum", args=[dy, out, M, C, C, C, E_F32, E_F32, None, 0, 0, db2, torch.zeros(64 * C)])
    assert torch.equal(out, dy) and torch.allclose(db2, b2.grad, atol=1e-5)
    # with DropPath scales: rows of sample q are scaled by rowscale[q, row >= n_prompt]
    rs = torch.tensor([[0.5, 2.0], [0.0, 1.5]])
    E.call("rowscale_cast_colsum", args=[dy, out, M, C, C, C, E_F32, E_F32, rs, 20, 3, db2, torch.zeros(64 * C)])
    scale = torch.cat([rs[q, (torch.arange(20) >= 3).long()] for q in range(2)])
    assert torch.allclose(out, dy * scale[:, None]) and torch.allclose(db2, (dy * scale[:, None]).sum(0), atol=1e-5)


def test_segcopy_and_ctr_mix_bf16_output():
    """mtt_segcopy against plain strided copies (vector, scalar, transposing and split segments through raw addresses), and the
    non-accumulating cross-task mix written in bf16."""
    import numpy as np
    g = g_(8)
    src = torch.randn(6, 20, generator=g)
    dst = torch.full((6, 24), 7.0, dtype=torch.bfloat16)
    hi, lo = torch.zeros(6, 24, dtype=torch.bfloat16), torch.zeros(6, 24, dtype=torch.bfloat16)
    tr = torch.full((20, 6), 7.0)
    rows = [[src.data_ptr(), dst.data_ptr(), 0, 120, 6, 20, 0, 20, 1, 0, 24, 1, 0, 1, 1, 0],             # fp32 -> bf16 with padding, "vector"
            [src.data_ptr(), hi.data_ptr(), lo.data_ptr(), 120, 6, 20, 0, 20, 1, 0, 24, 1, 0, 2, 0, 0],   # fp32 -> hi / lo planes
            [src.data_ptr(), tr.data_ptr(), 0, 4096, 20, 6, 0, 1, 20, 0, 6, 1, 0, 0, 0, 1]]               # transposing: box (1, 20, 6), 1 tile slot
    table = torch.from_numpy(np.asarray(rows, dtype=np.int64))
    E.call("segcopy", table=table, chunk_seg=torch.tensor([0, 1, 2], dtype=torch.int32), chunk_off=torch.zeros(3, dtype=torch.int64), n_chunks=3,
           src_base=0, dst_base=0)
    assert torch.equal(dst[:, :20], src.to(torch.bfloat16)) and float((dst[:, 20:].float() - 7.0).abs().max()) == 0.0     # padding untouched
    assert torch.equal(hi[:, :20], src.to(torch.bfloat16)) and torch.allclose(hi[:, :20].float() + lo[:, :20].float(), src, atol=1e-4)
    assert torch.equal(tr, src.t().contiguous())
    T, B, rpb, ld, C = 3, 2, 5, 16, 12
    fea = torch.randn(T, B * rpb, ld, generator=g)
    fea[..., C:] = 0
    wm = torch.randn(B, T, T, generator=g)
    out = torch.full((T, B * rpb, ld), 3.0, dtype=torch.bfloat16)
    E.call("ctr_mix", fea=fea, out=out, wmix=wm, T=T, B=B, rows_per_b=rpb, ld=ld, C=C, fea_dtype=E_F32, accumulate=0, out_dtype=1)
    ref = torch.einsum("bts,sbrc->tbrc", wm, fea.view(T, B, rpb, ld)).reshape(T, B * rpb, ld)
    assert torch.allclose(out.float(), ref, atol=2e-2, rtol=1e-2)
