"""-m gpu: BASELINE.json's full-size configuration (TaskPrompter ViT-L/16, 512x512, 6 tasks, N = 1030 tokens, 16 heads).

Element-wise against the CPU oracle / the fp64 ABI emulator at the benchmarked shape:
  * one training step in the BENCHMARKED mode (bf16: flash attention forward + backward, LDS-DMA GEMMs, implicit-GEMM convs):
    outputs and every parameter gradient vs the oracle's autograd (B = 2),
  * the flash attention kernels alone at (B = 2, N = 1030, nH = 16, T = 6) — eight 128-row tiles + the 6-row tail, XCD remap — and
    at N = 8194 (cfg5's sequence length), forward and backward, vs the fp64 emulator; plus an online-softmax spike in the tail tile,
  * the LDS-DMA bf16 GEMM / conv kernels vs the independently parity-gated x3 kernels at the bench's shapes.
(The eval forward of every BASELINE config vs the oracle is in test_gpu_configs.py.)
Size-independent properties:
  * eval-mode batch invariance: an image's prediction does not depend on what else is in the batch,
  * the hand-written backward is the derivative of the forward: a central finite difference of the real MultiTaskLoss along a
    random parameter direction matches <grad, direction> (x3 mode, train-mode BatchNorm, DropPath off).
"""
import pytest
import torch

import conftest
import parity_util as pu


def _build(prec, seed=0):
    import mtt_amd
    p = mtt_amd.factory.make_p(mtt_amd.factory.TASK_ORDER, (512, 512), backbone="TaskPrompter_vitL", head="conv", embed_dim=300,
                               final_embed_dim=350, chan_nheads=1, use_ctr=True, prec=prec, drop_path_rate=0.0)
    torch.manual_seed(seed)
    return p, mtt_amd.factory.get_model(p).cuda()


def _rel(a, b):
    return float((a.float() - b.float()).norm() / b.float().norm())


@pytest.mark.gpu
def test_fullsize_eval_batch_invariance_and_mode_agreement():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    p, m16 = _build("bf16")
    _, m32 = _build("x3")
    m32.load_state_dict(m16.state_dict())
    m16.eval(), m32.eval()
    x = torch.randn(3, 3, 512, 512, generator=torch.Generator().manual_seed(1)).cuda()
    with torch.no_grad():
        all16, one16 = m16(x), m16(x[1:2])
        all32, one32 = m32(x[:2]), m32(x[1:2])
    for t in p.TASKS.NAMES:
        assert all16[t].shape[0] == 3 and all16[t].shape[-2:] == (512, 512)
        assert torch.isfinite(all16[t]).all()
        assert _rel(all16[t][1:2], one16[t]) < 1e-2, t          # bf16: the GEMM variant (tile, split) depends on the row count -> rounding order
        assert _rel(all32[t][1:2], one32[t]) < 1e-4, t
        assert _rel(all16[t][:2], all32[t]) < 6e-2, (t, _rel(all16[t][:2], all32[t]))


@pytest.mark.gpu
def test_fullsize_backward_is_the_derivative_of_forward():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import mtt_amd
    from oracle import losses_oracle
    p, model = _build("x3")
    model.train()
    crit = losses_oracle.MultiTaskLoss(p, p.TASKS.NAMES).cuda()
    x = torch.randn(2, 3, 512, 512, generator=torch.Generator().manual_seed(2)).cuda()
    gt = mtt_amd.losses.synthetic_targets(p, 2, 512, 512, "cuda", seed=3)
    params = [q for q in model.parameters() if q.requires_grad]

    def loss():
        return crit(model(x), gt)["total"]

    base = loss()
    base.backward()
    g = torch.Generator(device="cuda").manual_seed(5)
    dirs = []
    for q in params:                      # per parameter: (unit gradient + unit random direction) scaled to the parameter's norm
        r = torch.randn(q.shape, device="cuda", generator=g)
        d = q.grad / (q.grad.norm() + 1e-30) + r / r.norm()
        dirs.append(d * q.detach().norm().clamp_min(1e-2))
    analytic = float(sum((q.grad.double() * d.double()).sum() for q, d in zip(params, dirs)))
    assert all(q.grad is not None and torch.isfinite(q.grad).all() for q in params)
    eps = 0.025 / abs(analytic)          # first-order loss change of +-0.025 (loss ~ 37): far above fp32 noise, inside the linear regime

    def shifted(sign):
        with torch.no_grad():
            for q, d in zip(params, dirs):
                q.add_(d, alpha=sign * eps)
            v = float(loss().double())
            for q, d in zip(params, dirs):
                q.add_(d, alpha=-sign * eps)
        return v

    numeric = (shifted(+1) - shifted(-1)) / (2 * eps)
    assert abs(numeric - analytic) <= 3e-2 * max(abs(analytic), abs(numeric)) + 1e-4, (numeric, analytic, float(base.detach()))


@pytest.mark.gpu
def test_ns6_training_step_bf16_matches_oracle_autograd():
    """The benchmarked mode end to end at the benchmarked shape: train-mode forward (batch-statistic BN) + hand-written backward
    (flash attention backward, fast256 dgrad / transposed wgrad, conv dgrad / wgrad) vs the oracle's autograd on the host."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import time
    import train_check
    from oracle import taskprompter_oracle as tpo
    from tests.golden.make_golden import loss_of
    cfg, sd, x, _ = pu.oracle_eval("ns6", 2)
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items() if v.dtype.is_floating_point and "running_" not in k}
    t0 = time.time()
    ref_out = tpo.forward(dict(sd, **params), cfg, x, training=True)
    loss_of(ref_out).backward()
    pu.report("oracle_time", config="ns6 train fwd+bwd", batch=2, seconds=round(time.time() - t0, 1))
    # x3f: x3 forward (north_star's 1e-3) with the bf16 backward on the hi planes — gradient error is bf16-class, from exact activations
    for prec, ftol, mtol in (("bf16", 4e-2, 6e-2), ("x3", 1e-3, 2e-3), ("x3f", 1e-3, 6e-2)):
        model = conftest.build_product_model(cfg, prec, "cuda")
        model.load_state_dict(sd, strict=True)
        model.train()
        out = model(x.cuda())
        loss_of({k: v.cpu() for k, v in out.items()}).backward()
        fwd = {t: pu.rel(out[t].detach(), ref_out[t].detach()) for t in ref_out}
        rels, errs = [], {}
        for k, prm in model.named_parameters():
            ref = params[k].grad if params[k].grad is not None else torch.zeros_like(params[k])
            assert prm.grad is not None and torch.isfinite(prm.grad).all(), k
            errs[k] = train_check.grad_err(prm.grad, ref)
            if errs[k].ref > 1e-6:
                rels.append((errs[k].err / errs[k].ref, k))
        rels.sort(reverse=True)
        med = rels[len(rels) // 2][0]
        bad, checked, below = train_check.per_param_violations(errs, "x3f" if prec != "bf16" else "bf16")
        pu.report("train_parity", config="ns6", batch=2, prec=prec, fwd_worst=max(fwd.values()), grad_median=med, grad_worst=rels[0][0],
                  grad_worst_param=rels[0][1], grad_p90=rels[len(rels) // 10][0], per_head=fwd, per_param_checked=checked,
                  per_param_below_floor=below, per_param_violations=len(bad), worst_cos=min(v.cos for v in errs.values() if v.numel >= 8 and v.ref > 1e-6))
        assert max(fwd.values()) < ftol, fwd
        assert med < mtol, (med, rels[:3])
        # every parameter of the benchmarked model, not the median (bounds and calibration: train_check.PER_PARAM)
        train_check.assert_per_param(errs, "x3f" if prec != "bf16" else "bf16")
        del model, out
        torch.cuda.empty_cache()


@pytest.mark.gpu
def test_cfg4_invpt_training_step_matches_oracle_autograd():
    """BASELINE's 8-GPU config (InvPT ViT-L, 6 tasks, 512x512) at FULL size: train-mode forward (batch-statistic BatchNorm, intermediate
    supervision outputs) + the hand-written backward of the ViT encoder and the InvPT decoder vs the oracle's autograd (B = 1).
    x3: fp32-class; bf16 / x3f: bf16-class gradients (x3f: from the x3 forward).  Parameters the reference leaves without a gradient
    (find_unused_parameters = True) must get none here either."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import time
    from oracle import invpt_oracle as ipo
    from tests.golden.make_golden import loss_of
    cfg, sd, x, _ = pu.oracle_eval("cfg4_6", 1)
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items() if v.dtype.is_floating_point and "running_" not in k}
    t0 = time.time()
    ref_out = ipo.forward(dict(sd, **params), cfg, x, training=True)
    loss_of(ref_out).backward()
    pu.report("oracle_time", config="cfg4_6 train fwd+bwd", batch=1, seconds=round(time.time() - t0, 1))
    dead_ref = sorted(k for k, v in params.items() if v.grad is None)
    # x3 gradient bound 2e-2 (measured 5.6e-3 median, 1.1e-2 worst at B = 1): the decoder is ReLU + train-mode BatchNorm over 256-pixel
    # maps at this batch; the fp64 emulator of the same descriptors agrees with the oracle's autograd to 8e-6 (tests/test_host_cpu.py),
    # so what remains on the GPU is fp32 summation order through those layers, not a wiring difference.  The forward bound stays 1e-3.
    for prec, ftol, mtol in (("x3", 1e-3, 2e-2), ("x3f", 1e-3, 2e-1), ("bf16", 4e-2, 2e-1)):
        model = conftest.build_product_model(cfg, prec, "cuda")
        model.load_state_dict(sd, strict=True)
        model.train()
        out = model(x.cuda())
        cpu = {k: v.cpu() for k, v in out.items() if k != "inter_preds"}
        cpu["inter_preds"] = {k: v.cpu() for k, v in out["inter_preds"].items()}
        loss_of(cpu).backward()
        fwd = {t: pu.rel(cpu[t].detach(), ref_out[t].detach()) for t in cpu if t != "inter_preds"}
        fwd.update({"inter/" + t: pu.rel(cpu["inter_preds"][t].detach(), v.detach()) for t, v in ref_out["inter_preds"].items()})
        rels, dead = [], []
        for k, prm in model.named_parameters():
            if params[k].grad is None:
                dead.append(k)
                assert prm.grad is None or float(prm.grad.abs().max()) == 0.0, k
                continue
            assert prm.grad is not None and torch.isfinite(prm.grad).all(), k
            n = float(params[k].grad.norm())
            if n > 1e-6:
                rels.append((float((prm.grad.cpu() - params[k].grad).norm()) / n, k))
        rels.sort(reverse=True)
        med = rels[len(rels) // 2][0]
        pu.report("train_parity", config="cfg4_6", batch=1, prec=prec, fwd_worst=max(fwd.values()), grad_median=med, grad_worst=rels[0][0],
                  grad_worst_param=rels[0][1], grad_p90=rels[len(rels) // 10][0], dead_parameters=len(dead))
        assert sorted(dead) == dead_ref
        assert max(fwd.values()) < ftol, fwd
        assert med < mtol, (med, rels[:3])
        del model, out
        torch.cuda.empty_cache()


def _attn_inputs(B, N, nH, seed, spike=None):
    g = torch.Generator().manual_seed(seed)
    C = nH * 64
    qkv = torch.randn(B * N, 3 * C, generator=g)
    if spike is not None:                      # one key row that dominates one query row's scores: the running max jumps at that tile
        qrow, krow = spike
        qkv[krow, C:C + 64] = qkv[qrow, :64] * 6.0
    return qkv.to(torch.bfloat16)


@pytest.mark.gpu
@pytest.mark.parametrize("B,N,nH,T,spike", [(2, 1030, 16, 6, None), (1, 1030, 2, 6, (1027, 1029)), (1, 8194, 2, 2, None)],
                         ids=["ns6_shape", "ns6_tail_spike", "cfg5_seq"])
def test_flash_attention_kernels_at_full_size(B, N, nH, T, spike):
    """mtt_attn_fwd (swapped-product flash kernel) and mtt_attn_bwd vs the fp64 emulator at the benchmarked sequence lengths."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import gpu_cases
    from oracle import abi_emul
    C = nH * 64
    qkv = _attn_inputs(B, N, nH, 31, spike)
    fw = dict(qkv=qkv, out=torch.zeros(B * N, C, dtype=torch.bfloat16), rawlog=torch.zeros(B, nH, T, N), lse=torch.zeros(B, nH, N),
              B=B, N=N, nH=nH, T=T, dtype=1, prec=0, scale=0.125)
    r = gpu_cases.run_case("attn_fwd", fw, None, dict(f32=2e-3, bf16=6e-3))
    pu.report("kernel_parity", kernel="attn_fwd_fast", B=B, N=N, nH=nH, T=T, errs={k: v[0] if isinstance(v, tuple) else v for k, v in r["errs"].items()})
    assert r["ok"], r["errs"]
    g = torch.Generator().manual_seed(32)
    kw = dict(qkv=qkv, out=fw["out"], rawlog=None, lse=fw["lse"], B=B, N=N, nH=nH, T=T, dtype=1, prec=0, scale=0.125,
              xargs=[torch.randn(B * N, C, generator=g).to(torch.bfloat16), torch.randn(B, nH, T, N, generator=g) * 0.05,
                     torch.zeros(B * N, 3 * C, dtype=torch.bfloat16), torch.zeros(B, nH, 2, (N + 3) // 4 * 4)])
    r = gpu_cases.run_case("attn_bwd", kw, None, dict(f32=5e-3, bf16=1.5e-2))
    pu.report("kernel_parity", kernel="attn_bwd", B=B, N=N, nH=nH, T=T, errs={k: v[0] if isinstance(v, tuple) else v for k, v in r["errs"].items()})
    assert r["ok"], r["errs"]
    # the x3 flash kernel on split planes (forward of the x3f mode) on fp32 inputs that are not bf16-representable: fp32-class everywhere
    gq = torch.Generator().manual_seed(33)
    q32 = torch.randn(B * N, 3 * C, generator=gq)
    if spike is not None:
        q32[spike[1], C:C + 64] = q32[spike[0], :64] * 6.0
    qh = q32.to(torch.bfloat16)
    sx = dict(qkv=qh, qkv_lo=(q32 - qh.float()).to(torch.bfloat16), out=torch.zeros(B * N, C, dtype=torch.bfloat16),
              out_lo=torch.zeros(B * N, C, dtype=torch.bfloat16), rawlog=torch.zeros(B, nH, T, N), lse=torch.zeros(B, nH, N),
              B=B, N=N, nH=nH, T=T, dtype=2, prec=1, scale=0.125)
    r = gpu_cases.run_case("attn_fwd", sx, None, dict(f32=3e-5, bf16=6e-3, split=3e-5, split_pairs=[("out", "out_lo")]))
    pu.report("kernel_parity", kernel="attn_fwd_x3_split", B=B, N=N, nH=nH, T=T, errs={k: v[0] if isinstance(v, tuple) else v for k, v in r["errs"].items()})
    assert r["ok"], r["errs"]


@pytest.mark.gpu
def test_dma_gemm_and_conv_kernels_agree_with_x3_at_bench_shapes():
    """The bf16 LDS-DMA GEMM (fast256) and the bf16 implicit-GEMM conv at the shapes of the benchmarked step against the x3 kernels
    (fp32-class, parity-gated against the emulator and the oracle) on the same bf16-representable operands: what remains is fp32
    accumulation order and the bf16 rounding of the output."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import mtt_amd
    ops = mtt_amd.ops
    bf, x3 = ops.Prec("bf16"), ops.Prec("x3")
    g = torch.Generator(device="cuda").manual_seed(3)
    for tag, M, N, K in (("qkv", 8 * 1030, 3072, 1024), ("fc1", 8 * 1030, 4096, 1024), ("fc2", 8 * 1030, 1024, 4096), ("proj", 8 * 1030, 1024, 1024)):
        x = torch.randn(M, K, device="cuda", generator=g).bfloat16()
        w = (torch.randn(1, N, K, device="cuda", generator=g) / K ** 0.5).bfloat16()
        b = torch.randn(1, N, device="cuda", generator=g)
        assert mtt_amd._lib.gemm_variant(A=x, B=w, D=x, M=M, N=N, K=K, a_dtype=1, b_dtype=1, d_dtype=1, prec=0, lda=K, ldb=K, ldd=N, batch=1) == 3
        y16 = ops.linear(x, w, N, bf, bias=b, out_dtype=torch.float32)
        y32 = ops.linear(x.float(), w.float(), N, x3, bias=b)
        e = pu.rel(y16, y32)
        pu.report("kernel_parity", kernel="gemm_fast256", shape=tag, M=M, N=N, K=K, rel=e)
        assert e < 2e-5, (tag, e)
        # the same kernel on MTT_SPLIT planes (K-concatenated x3 product) with fp32 operands that are NOT bf16-representable
        xf = torch.randn(M, K, device="cuda", generator=g)
        wp = torch.nn.Parameter(torch.randn(N, K, device="cuda", generator=g) / K ** 0.5)
        ws = ops.pack_linear_split([wp], ("t_split", tag))
        assert mtt_amd._lib.gemm_variant(A=x, B=w, D=x, A_lo=x, B_lo=w, M=M, N=N, K=K, a_dtype=2, b_dtype=2, d_dtype=0, prec=1, lda=K, ldb=K, ldd=N,
                                         batch=1) == 8
        ys = ops.linear(ops.split_cast(xf), ws, N, x3, bias=b, out_dtype=torch.float32)
        yr = ops.linear(xf, wp.detach()[None].contiguous(), N, x3, bias=b)
        e = pu.rel(ys, yr)
        pu.report("kernel_parity", kernel="gemm_dma_split", shape=tag, M=M, N=N, K=K, rel=e)
        assert e < 3e-5, (tag, e)
    B, H, W, F = 4, 128, 128, 350
    Fp = ops.pad8(F)
    xin = torch.zeros(2, B * H * W, Fp, device="cuda")
    xin[..., :F] = torch.randn(2, B * H * W, F, device="cuda", generator=g)
    xin = xin.bfloat16()
    ws = [torch.nn.Parameter(torch.randn(F, F, 3, 3, device="cuda", generator=g) / (9 * F) ** 0.5) for _ in range(2)]
    for wt in ws:
        wt.data = wt.data.bfloat16().float()
    y16 = ops.conv3x3(xin, ops.pack_conv3(ws, bf, "t16"), F, F, B, H, W, bf, out_dtype=torch.float32)
    y32 = ops.conv3x3(xin.float(), ops.pack_conv3(ws, x3, "t32"), F, F, B, H, W, x3)
    e = pu.rel(y16[..., :F], y32[..., :F])
    pu.report("kernel_parity", kernel="conv3x3_bf16", shape="head conv 350ch 128x128", rel=e)
    assert e < 2e-5, e


@pytest.mark.gpu
@pytest.mark.timeout(2400)
def test_swinb_fullsize_training_step_matches_oracle_autograd():
    """TaskPrompter Swin-B at cs_swinB's FULL size (taskprompter_swin.py:120-774; cs_swinB_taskprompter.yml: 1024 x 2048 x 0.75, window 12,
    depths 2-2-18-2, DEConvHead; B = 1): train-mode forward + the hand-written backward (shifted-window attention with prompt rows, channel
    attention, patch merging of features / prompts / attention maps, decoder) vs the oracle's autograd.  Rounds 2-3 checked these gradients on
    two miniatures only.  x3: fp32-class; bf16: bf16-class.  Parameters the reference leaves without a gradient must get none here."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import time
    import train_check
    cfg = pu.get_cfg("cs_swinB")
    contract = pu.contract_of(cfg)
    torch.set_num_threads(conftest.HOST_THREADS)
    cache = {}
    for prec, ftol, mtol in (("x3", 1e-3, 3e-3), ("bf16", 5e-2, 1.5e-1)):
        t0 = time.time()
        fwd, errs, dead = train_check.swin_grad_errors("cs_swinB", prec, "cuda", batch=1, contract=contract, ref_cache=cache)
        worst, med = train_check.summarize(errs)
        rels = sorted((v.err / max(v.ref, 1e-30) for v in errs.values() if v.ref > 1e-6), reverse=True)
        pu.report("train_parity", config="cs_swinB", batch=1, prec=prec, fwd_worst=max(fwd.values()), grad_median=med, grad_worst=worst[0],
                  grad_worst_param=worst[1], grad_p90=rels[len(rels) // 10], n_params=len(errs), n_dead=len(dead), per_head=fwd,
                  seconds=round(time.time() - t0, 1))
        assert max(fwd.values()) < ftol, fwd
        assert med < mtol, (med, worst)
        torch.cuda.empty_cache()
