"""CPU (`-m "not gpu"`): host-side logic of the product.
  * libmtt_hip.so loads and exports every symbol include/mtt_hip.h declares; ctypes mirrors match sizeof()
  * state-dict contract == the reference's (names, shapes, order) -> load_state_dict(strict=True) both ways
  * the nn.Module wiring (descriptor construction, strides, padding) on the ABI emulator reproduces the
    golden outputs of the unmodified reference
  * the product refuses to run its kernels on CPU tensors (no fallback)"""
import math
import os
import re

import pytest
import torch

import conftest
from oracle import configs, losses_oracle, weights


def test_library_loads_and_exports_header_symbols():
    import mtt_amd
    lib = mtt_amd._lib.load()
    header = open(os.path.join(conftest.ROOT, "include", "mtt_hip.h")).read()
    declared = set(re.findall(r"\b(mtt_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations parsed"
    for sym in sorted(declared):
        assert hasattr(lib, sym), f"{sym} declared in mtt_hip.h but not exported"
    assert set(mtt_amd._lib.EXPORTS) <= declared


def test_no_cpu_fallback():
    import mtt_amd
    x = torch.zeros(8, 8)
    with pytest.raises(RuntimeError):
        mtt_amd._lib.call("cast2d", args=[x, x.clone(), 8, 8, 8, 8, 0, 0, 0])


@pytest.mark.parametrize("name", ["mini_ctr", "mini_win", "mini_deconv"])
def test_state_dict_contract_matches_reference(name):
    cfg = configs.taskprompter(name)
    meta, _ = conftest.load_golden(name)
    model = conftest.build_product_model(cfg, "x3")
    mine = [(k, list(v.shape)) for k, v in model.state_dict().items()]
    assert mine == [(k, list(s)) for k, s in meta["contract"]]
    model.load_state_dict(weights.synth_state_dict(meta["contract"], 0), strict=True)


@pytest.mark.parametrize("name", ["mini_ctr", "mini_win", "mini_deconv"])
@pytest.mark.parametrize("prec,tol", [("x3", 2e-5), ("x3f", 5e-5), ("bf16", 4e-2)])
def test_wiring_on_emulator_matches_golden(emulated, name, prec, tol):
    cfg = configs.taskprompter(name)
    meta, gold = conftest.load_golden(name)
    model = conftest.build_product_model(cfg, prec)
    model.load_state_dict(weights.synth_state_dict(meta["contract"], 0), strict=True)
    model.eval()
    x = weights.synth_images(meta["batch"], cfg["img_size"], 1)
    with torch.no_grad():
        out = model(x)
    for t, n in cfg["tasks"]:
        g = torch.from_numpy(gold[f"eval/{t}"])
        assert out[t].shape == g.shape and out[t].dtype == torch.float32
        assert float((out[t] - g).norm() / g.norm()) < tol, (t, prec)


@pytest.mark.parametrize("name", ["mini_ctr", "mini_deconv"])
def test_wiring_train_mode_batchnorm(emulated, name):
    cfg = configs.taskprompter(name)
    meta, gold = conftest.load_golden(name)
    model = conftest.build_product_model(cfg, "x3")
    model.load_state_dict(weights.synth_state_dict(meta["contract"], 0), strict=True)
    model.train()
    x = weights.synth_images(2, cfg["img_size"], 2)
    with torch.no_grad():
        out = model(x)
    for t, n in cfg["tasks"]:
        g = torch.from_numpy(gold[f"train/{t}"])
        assert float((out[t][:, :, ::2, ::2] - g).norm() / g.norm()) < 5e-5, t
    sd = model.state_dict()
    for k in sd:
        if k.endswith("running_mean") or k.endswith("running_var"):
            assert float((sd[k] - torch.from_numpy(gold[f"bn/{k}"])).abs().max()) < 1e-4, k


@pytest.mark.parametrize("name", ["mini_ctr", "mini_win", "mini_deconv"])
def test_training_gradients_on_emulator(emulated, name):
    """Every backward descriptor (dgrad/wgrad views, conv wgrad, attention bwd, modulation bwd, BN bwd, ...) on the
    ABI emulator vs the oracle's autograd: all parameter gradients within 1e-3 relative."""
    import train_check
    fwd, errs = train_check.grad_errors(name, "x3", "cpu")
    assert max(fwd.values()) < 5e-5
    worst, med = train_check.summarize(errs)
    assert worst[0] < 1e-3, worst


def test_wrapper_dd_label_map_size_on_the_emulator(emulated):
    """taskprompter_wrapper.py:17-27 through the product's wrapper and the ABI emulator: predictions at `p.dd_label_map_size` (40 x 56 from
    a 64 x 96 input) against the unmodified reference wrapper's fixture, and the gradients through that resize against the oracle's autograd
    (the -m gpu twin: tests/test_gpu_model.py::test_wrapper_dd_label_map_size_on_the_device)."""
    import numpy as np
    import train_check
    from oracle import taskprompter_oracle as tpo
    cfg = configs.taskprompter("mini_ctr_dd")
    meta, _ = conftest.load_golden("mini_ctr")
    gold = np.load(os.path.join(conftest.GOLDEN, "mini_ctr_dd.npz"))
    model = conftest.build_product_model(cfg, "x3", "cpu")
    assert tuple(model.target_size) == (40, 56)
    model.load_state_dict(weights.synth_state_dict(meta["contract"], 0), strict=True)
    model.eval()
    with torch.no_grad():
        out = model(weights.synth_images(2, cfg["img_size"], 1))
    for t, _ in cfg["tasks"]:
        g = torch.from_numpy(gold[f"eval/{t}"])
        assert out[t].shape == g.shape, (t, out[t].shape)
        assert float((out[t] - g).norm() / g.norm()) < 5e-5, t
    fwd, errs = train_check.grad_errors("mini_ctr_dd", "x3", "cpu")
    worst, med = train_check.summarize(errs)
    assert max(fwd.values()) < 5e-5 and worst[0] < 1e-2 and med < 1e-3, (fwd, worst, med)


def test_x3f_mode_forward_x3_on_split_planes_backward_bf16(emulated, monkeypatch):
    """x3f: the four encoder Linears of every block (and the attention between them) run on MTT_SPLIT planes in x3 arithmetic, the
    backward is the bf16 one (flash attention backward, bf16 weight / input gradients on the hi planes); outputs stay fp32-class,
    gradients bf16-accurate."""
    import mtt_amd
    import train_check
    seen = []
    inner = mtt_amd.ops.call

    def spy(name, **kw):
        seen.append((name, kw.get("a_dtype"), kw.get("prec"), kw.get("dtype")))
        return inner(name, **kw)
    monkeypatch.setattr(mtt_amd.ops, "call", spy)
    fwd, errs = train_check.grad_errors("mini_ctr", "x3f", "cpu")
    assert max(fwd.values()) < 5e-5, fwd
    worst, med = train_check.summarize(errs, floor=1e-4)
    assert med < 3e-2, (worst, med)
    train_check.assert_per_param(errs, "x3f")              # every parameter, not the median (the -m gpu twin asserts the same on the device)
    n_blocks = n_taps = 4
    # qkv, proj, fc1, fc2 per block + per tap fea_decode (on the planes `modulate` writes) and fea_fuse[0] (on the planes its epilogue writes)
    n_split = sum(1 for n, adt, pr, _ in seen if n == "gemm" and adt == 2 and pr == 1)
    # (with the multiple-of-32 pitch on the miniature's 52-channel maps — MTT_TEST_PITCH32_FROM — the decoder's convs and head GEMM join them)
    assert n_split == 4 * n_blocks + 2 * n_taps if mtt_amd.ops.pad8(52) == 56 else n_split > 4 * n_blocks + 2 * n_taps
    assert sum(1 for n, _, _, _ in seen if n == "modulate") == n_taps
    assert sum(1 for n, _, pr, dt in seen if n == "attn_fwd" and dt == 2 and pr == 1) == n_blocks
    assert sum(1 for n, *_ in seen if n == "attn_bwd") == n_blocks
    assert not any(n == "gemm" and pr == 1 for n, _, pr, _ in seen[len(seen) // 2 + 40:]), "x3 GEMMs in the backward half"


def test_x3f_decoder_convs_and_head_gemm_run_on_split_planes(emulated, monkeypatch):
    """Decoder widths that are multiples of 32 (352 / 1024 / 768 in the BASELINE configs; 64 here): in x3f the fea_fuse 3x3 convs take the
    split-plane implicit-GEMM kernel (mtt_gemm: MTT_OP_CONV_K on MTT_SPLIT operands, y0 written as planes by fea_fuse[0]'s epilogue) and the
    taps-first head GEMM runs on planes of the low-resolution task features — whole model on the ABI emulator vs the oracle's autograd."""
    import mtt_amd
    import train_check
    seen = []
    inner = mtt_amd.ops.call

    def spy(name, **kw):
        seen.append((name, kw.get("a_op"), kw.get("a_dtype"), kw.get("d_dtype"), kw.get("prec"), kw.get("N")))
        return inner(name, **kw)
    monkeypatch.setattr(mtt_amd.ops, "call", spy)
    fwd, errs = train_check.grad_errors("mini_p32", "x3f", "cpu")
    assert max(fwd.values()) < 5e-5, fwd
    worst, med = train_check.summarize(errs, floor=1e-4)
    assert med < 3e-2, (worst, med)
    train_check.assert_per_param(errs, "x3f")
    n_taps = 4
    convs = [s for s in seen if s[0] == "gemm" and s[1] == 2 and s[2] == 2 and s[4] == 1]             # OP_CONV_K on split planes, x3
    assert len(convs) == n_taps, len(convs)
    assert sum(1 for s in seen if s[0] == "gemm" and s[1] == 0 and s[2] == 2 and s[3] == 2 and s[5] == 64) == n_taps    # fea_fuse[0] -> planes
    assert sum(1 for s in seen if s[0] == "gemm" and s[1] == 0 and s[2] == 2 and s[5] == 9 * 64) == 1                   # the nine-tap head GEMM
    assert not any(s[0] == "gemm" and s[1] == 2 and s[2] == 0 and s[4] == 1 for s in seen), "a forward 3x3 conv still on the register-staged x3 kernel"


def test_split_plane_conv_chunks_the_batch_at_the_kernels_row_limit(emulated, monkeypatch):
    """mtt_gemm variant 9 addresses a batch member's pixel rows with 32-bit element offsets (M * pitch < 2^31, gemm_variant_for); beyond
    it ops.conv3x3 launches chunks of whole images instead of raising MTT_E_UNSUPPORTED.  With the limit lowered to one
    image the result must be bitwise the unchunked one; the pack falls back to the non-split layout outside the kernel's other limits."""
    import mtt_amd
    ops = mtt_amd.ops
    g = torch.Generator().manual_seed(0)
    Z, B, H, W, Ci, Co = 2, 3, 5, 4, 32, 24
    prec = ops.Prec("x3f")
    ws = [torch.nn.Parameter(torch.randn(Co, Ci, 3, 3, generator=g) * 0.1) for _ in range(Z)]
    wp = ops.pack_conv3(ws, prec, "chunk_test")
    assert isinstance(wp, ops.Split)
    x = torch.randn(Z, B * H * W, Ci, generator=g)
    bias = torch.randn(Z, Co, generator=g)
    whole = ops.conv3x3(x, wp, Co, Ci, B, H, W, prec, bias=bias, out_dtype=torch.float32)
    calls = []
    inner = ops.call
    monkeypatch.setattr(ops, "call", lambda name, **kw: (calls.append((name, kw.get("M"))), inner(name, **kw))[1])
    monkeypatch.setattr(ops, "SPLIT_CONV_MAX_ELEMS", H * W * Ci)
    chunked = ops.conv3x3(x, wp, Co, Ci, B, H, W, prec, bias=bias, out_dtype=torch.float32)
    assert [c for c in calls if c[0] == "gemm"] == [("gemm", H * W)] * B
    assert torch.equal(whole, chunked)
    ref = torch.nn.functional.conv2d(x[0].view(B, H, W, Ci).permute(0, 3, 1, 2), ws[0].detach(), bias[0], padding=1).permute(0, 2, 3, 1)
    assert float((whole[0].view(B, H, W, -1)[..., :Co] - ref).norm() / ref.norm()) < 1e-4
    assert ops.split_conv_ok(352, 352) and not ops.split_conv_ok(20) and not ops.split_conv_ok(8192, 64) and not ops.split_conv_ok(4096, 60000)
    assert ops.split_conv_ok(300) == (ops.pad8(300) == 320)        # the multiple-of-32 channel pitch from ops.PITCH32_FROM channels on
    ops.clear_pack_cache()


def test_conv_head_as_one_node_keeps_its_gradient_maps_in_the_backwards_dtype(emulated, monkeypatch):
    """ConvHeadFn (conv -> BatchNorm -> predictions as ONE autograd node): in x3f the head's gradient maps are bf16 tensors next to the
    fp32-stored conv output (mtt_bn_desc.g_dtype, the gather kernel's bf16 input) and the gradients equal the three-node form's within
    bf16 storage rounding; in x3 the two forms are the same arithmetic."""
    import importlib
    import mtt_amd
    import train_check
    ap = importlib.import_module("multi-task-transformer_amd.autograd_path")
    seen = []
    inner = mtt_amd.ops.call

    def spy(name, **kw):
        if name in ("bn_bwd_reduce", "bn_bwd_apply", "upconv4_gather"):
            seen.append((name, kw.get("g_dtype"), kw.get("y_dtype"), kw.get("z_dtype"), kw["dy"].dtype if "dy" in kw else None, kw["x"].dtype if "x" in kw else None))
        return inner(name, **kw)
    monkeypatch.setattr(mtt_amd.ops, "call", spy)
    _, fused = train_check.grad_errors("mini_ctr", "x3f", "cpu")
    head = [s_ for s_ in seen if s_[0] == "bn_bwd_reduce"][0]            # the first BatchNorm backward of the step is the heads'
    assert head[1] == 2 and head[4] == torch.bfloat16 and head[5] == torch.float32, head
    gath = [s_ for s_ in seen if s_[0] == "upconv4_gather"][0]
    assert gath[2] == 1 and gath[3] == 1, gath                             # bf16 in, bf16 out: no cast pass before the tap GEMMs
    monkeypatch.setattr(ap, "FUSE_HEAD_NODE", False)
    seen.clear()
    _, plain = train_check.grad_errors("mini_ctr", "x3f", "cpu")
    assert [s_ for s_ in seen if s_[0] == "bn_bwd_reduce"][0][4] == torch.float32
    train_check.assert_per_param(fused, "x3f")
    train_check.assert_per_param(plain, "x3f")
    wf, mf = train_check.summarize(fused, floor=1e-4)
    wp, mp = train_check.summarize(plain, floor=1e-4)
    assert mf < 3e-2 and mf < 1.5 * mp + 1e-3, (mf, mp)                    # against the oracle's autograd: the same bf16-class accuracy
    for prec_name in ("x3",):
        monkeypatch.setattr(ap, "FUSE_HEAD_NODE", True)
        _, a = train_check.grad_errors("mini_ctr", prec_name, "cpu")
        monkeypatch.setattr(ap, "FUSE_HEAD_NODE", False)
        _, b = train_check.grad_errors("mini_ctr", prec_name, "cpu")
        assert all(abs(a[k][0] - b[k][0]) <= 1e-6 * max(a[k][1], 1e-12) + 1e-12 for k in a), "x3: the node fusion must not change the arithmetic"


@pytest.mark.parametrize("prec,ftol,gtol", [("x3", 5e-5, 1e-3), ("x3f", 5e-5, None)])
def test_non_tap_blocks_skip_their_dead_side_channels(emulated, monkeypatch, prec, ftol, gtol):
    """Six blocks of which two are not taps (as 20 of ViT-L's 24): their channel-logit pass is not launched and their rawlog / rawchan
    gradients arrive as None (set_materialize_grads(False)) — outputs and every gradient still match the oracle, which computes the
    reference's full graph (taskprompter.py:216-250 for every block)."""
    import mtt_amd
    import train_check
    seen = []
    inner = mtt_amd.ops.call
    monkeypatch.setattr(mtt_amd.ops, "call", lambda name, **kw: (seen.append(name), inner(name, **kw))[1])
    fwd, errs = train_check.grad_errors("mini_skip", prec, "cpu")
    assert max(fwd.values()) < ftol, fwd
    assert seen.count("chan_logits") == 4 and seen.count("chan_logits_bwd") == 4          # taps (blocks 1, 3, 4) + the last block; not 6
    if gtol is not None:
        worst, med = train_check.summarize(errs)
        assert worst[0] < 1e-2 and med < gtol, (worst, med)
    else:
        train_check.assert_per_param(errs, "x3f")
    model = conftest.build_product_model(configs.taskprompter("mini_skip"), prec, "cpu").eval()
    seen.clear()
    with torch.no_grad():
        model(weights.synth_images(1, (64, 96), 3))
    assert seen.count("chan_logits") == 4


def test_fused_head_and_fuse_tail_nodes_with_frozen_inputs(emulated):
    """ConvHeadFn / FuseTailFn run other Functions' forward / backward as STAGES with a stand-in ctx: with part of their inputs
    frozen (requires_grad = False: head convs, one task's fea_fuse tail, every BatchNorm affine) the step must still run, frozen tensors
    get no gradient and every other parameter gets exactly the gradient of the all-trainable run."""
    import mtt_amd
    from oracle import taskprompter_oracle as tpo  # noqa: F401
    from tests.golden.make_golden import loss_of
    cfg = configs.taskprompter("mini_ctr")
    meta, _ = conftest.load_golden("mini_ctr")
    sd = weights.synth_state_dict(meta["contract"], 0)
    x = weights.synth_images(2, cfg["img_size"], 2)

    def run(freeze):
        model = conftest.build_product_model(cfg, "x3f", "cpu")
        model.load_state_dict(sd, strict=True)
        model.train()
        frozen = [k for k, q in model.named_parameters() if freeze(k)]
        for k, q in model.named_parameters():
            q.requires_grad_(k not in frozen)
        loss_of(model(x)).backward()
        return {k: (None if q.grad is None else q.grad.clone()) for k, q in model.named_parameters()}, frozen

    full, _ = run(lambda k: False)
    part, frozen = run(lambda k: "mt_proj.0" in k or ".fea_fuse.1.depth." in k or ".mt_proj.1." in k or "fea_fuse.2.edge.2." in k)
    assert len(frozen) > 10
    for k, gfull in full.items():
        if k in frozen:
            assert part[k] is None, k
        else:
            assert part[k] is not None and torch.equal(part[k], gfull), k


def test_conv_head_prologue_form_matches_the_three_stage_form(emulated, monkeypatch):
    """ConvHeadFn in x3f training (round 6): BatchNorm + GELU applied while the prediction GEMM loads its operand (mtt_gemm_desc.a_scale),
    no bn_apply launch and no fp32 activated map for the heads; the backward reads the bf16 side copy.  Predictions equal the three-stage
    form's to fp32 rounding, every gradient stays inside the x3f per-parameter bound, BatchNorm running statistics are updated alike."""
    import importlib
    import mtt_amd
    import train_check
    ap = importlib.import_module("multi-task-transformer_amd.autograd_path")
    seen = []
    inner = mtt_amd.ops.call

    def spy(name, **kw):
        seen.append((name, kw.get("a_scale") is not None, kw.get("rows")))
        return inner(name, **kw)
    monkeypatch.setattr(mtt_amd.ops, "call", spy)
    cfg = configs.taskprompter("mini_ctr")
    meta, _ = conftest.load_golden("mini_ctr")
    sd = weights.synth_state_dict(meta["contract"], 0)
    x = weights.synth_images(2, cfg["img_size"], 2)

    def run(flag):
        monkeypatch.setattr(ap, "HEAD_PROLOGUE", flag)
        seen.clear()
        model = conftest.build_product_model(cfg, "x3f", "cpu")
        model.load_state_dict(sd, strict=True)
        model.train()
        out = model(x)
        calls = list(seen)
        return out, {k: v.clone() for k, v in model.state_dict().items() if "running_" in k}, calls

    of, bf, cf = run(True)
    op, bp, cp = run(False)
    T = len(cfg["tasks"])
    assert sum(1 for n, pro, _ in cf if n == "gemm" and pro) == T and not any(pro for _, pro, _ in cp)
    head_rows = 2 * 16 * cfg["img_size"][0] // 16 * cfg["img_size"][1] // 16
    assert sum(1 for n, _, r in cp if n == "bn_apply" and r == head_rows) == 1 and not any(n == "bn_apply" and r == head_rows for n, _, r in cf)
    for t in of:
        assert float((of[t].detach() - op[t].detach()).norm() / op[t].detach().norm()) < 2e-6, t
    for k in bf:
        assert torch.equal(bf[k], bp[k]), k
    monkeypatch.setattr(ap, "HEAD_PROLOGUE", True)
    _, errs = train_check.grad_errors("mini_ctr", "x3f", "cpu")
    train_check.assert_per_param(errs, "x3f")


def test_bf16_training_uses_flash_attention_backward(emulated, monkeypatch):
    """bf16 mode routes the attention backward to mtt_attn_bwd (flash, no N x N buffer); gradients stay bf16-accurate."""
    import mtt_amd
    import train_check
    seen = []
    inner = mtt_amd.ops.call
    monkeypatch.setattr(mtt_amd.ops, "call", lambda name, **kw: (seen.append(name), inner(name, **kw))[1])
    fwd, errs = train_check.grad_errors("mini_ctr", "bf16", "cpu")
    assert "attn_bwd" in seen and "softmax_bwd" not in seen
    worst, med = train_check.summarize(errs, floor=1e-4)
    assert max(fwd.values()) < 4e-2 and med < 6e-2, (worst, med)


def test_training_with_injected_droppath_masks(emulated):
    import train_check
    g = torch.Generator().manual_seed(3)
    drop = [(torch.bernoulli(torch.full((4, 2), 0.6), generator=g) / 0.6) for _ in range(4)]
    fwd, errs = train_check.grad_errors("mini_ctr", "x3", "cpu", drop=drop)
    assert max(fwd.values()) < 5e-5
    worst, med = train_check.summarize(errs)
    assert worst[0] < 1e-3, worst


def _invpt_outputs_vs_golden(model, cfg, meta, gold, tol, device="cpu"):
    es = meta["eval_stride"]
    model.eval()
    with torch.no_grad():
        out = model(weights.synth_images(meta["batch"], cfg["img_size"], 1).to(device))
    for t, _ in cfg["tasks"]:
        for key, val in ((f"eval/{t}", out[t]), (f"eval/inter/{t}", out["inter_preds"][t])):
            g = torch.from_numpy(gold[key])
            assert float((val.cpu()[:, :, ::es, ::es] - g).norm() / g.norm()) < tol, key
    model.train()
    with torch.no_grad():
        out = model(weights.synth_images(2, cfg["img_size"], 2).to(device))
    for t, _ in cfg["tasks"]:
        g = torch.from_numpy(gold[f"train/{t}"])
        assert float((out[t].cpu()[:, :, ::2, ::2] - g).norm() / g.norm()) < tol * 2, t


@pytest.mark.parametrize("prec,tol", [("x3", 2e-5), ("x3f", 2e-5), ("bf16", 4e-2)])
def test_invpt_contract_and_wiring_on_emulator(emulated, prec, tol):
    cfg = configs.invpt("mini")
    meta, gold = conftest.load_golden("mini")
    model = conftest.build_product_model(cfg, prec)
    assert [(k, list(v.shape)) for k, v in model.state_dict().items()] == [(k, list(s)) for k, s in meta["contract"]]
    model.load_state_dict(weights.synth_state_dict(meta["contract"], 0), strict=True)
    _invpt_outputs_vs_golden(model, cfg, meta, gold, tol)


def test_invpt_training_gradients_on_emulator(emulated):
    """InvPT training path (invpt_autograd.py): outputs and every parameter gradient vs the oracle's autograd."""
    import train_check
    fwd, errs, dead = train_check.invpt_grad_errors("mini8", "x3", "cpu")
    assert max(fwd.values()) < 5e-5, fwd
    worst, med = train_check.summarize(errs)
    assert worst[0] < 1e-3, worst
    assert any("scale_embed.2" in k for k in dead) and any("norm_mt." in k for k in dead)


def test_training_gradients_with_forced_split_k(emulated, monkeypatch):
    """Same gradient check with the split-K weight-gradient path forced on (production uses it for >= 4096-row reductions)."""
    import mtt_amd
    import train_check
    monkeypatch.setattr(mtt_amd.autograd_path if hasattr(mtt_amd, "autograd_path") else __import__("importlib").import_module(
        "multi-task-transformer_amd.autograd_path"), "SPLITK_MIN_ROWS", 8)
    ap = __import__("importlib").import_module("multi-task-transformer_amd.autograd_path")
    monkeypatch.setattr(ap, "SPLIT_ROW_UNIT", 8)             # task-batched decoder / 3x3-conv weight gradients: (task, slice) launches
    fwd, errs = train_check.grad_errors("mini_ctr", "x3", "cpu")
    worst, med = train_check.summarize(errs)
    assert worst[0] < 1e-3, worst


def test_fast_backward_gemm_routing_matches_general_path(emulated, monkeypatch):
    """bf16 mode: encoder dgrad / wgrad routed through transposed operands (fast GEMM path) give the same gradients as the
    general transposing-stager path (both on the emulator with bf16 operand rounding)."""
    import importlib
    import train_check
    ap = importlib.import_module("multi-task-transformer_amd.autograd_path")
    monkeypatch.setattr(ap, "FAST_BWD", False)
    _, ref = train_check.grad_errors("mini_ctr", "bf16", "cpu")
    monkeypatch.setattr(ap, "FAST_BWD", True)
    monkeypatch.setattr(ap, "FAST_MIN_DIM", 8)
    monkeypatch.setattr(ap, "FAST_MIN_ROWS", 8)
    _, got = train_check.grad_errors("mini_ctr", "bf16", "cpu")
    # grad_errors returns |g - oracle|; compare the two error profiles on the encoder weights
    for k in ref:
        if "blocks" in k and k.endswith("weight") and ref[k][1] > 1e-4:
            assert abs(got[k][0] - ref[k][0]) <= 0.5 * ref[k][0] + 1e-3 * ref[k][1], (k, got[k], ref[k])


def test_fused_clip_adam_matches_torch_adam_with_clipping(emulated, monkeypatch):
    """optim.FusedClipAdam (mtt_grad_sqnorm + mtt_adam_step on the emulator) vs clip_grad_norm_ + torch.optim.Adam, 3 steps, with a
    learning-rate schedule; state_dict layouts are interchangeable."""
    import mtt_amd
    monkeypatch.setattr(mtt_amd.ops, "adam_chunk", lambda: 65536)
    torch.manual_seed(0)
    shapes = [(70000,), (33, 17), (5,), (128, 64, 3)]
    ref = [torch.nn.Parameter(torch.randn(s)) for s in shapes]
    mine = [torch.nn.Parameter(q.detach().clone()) for q in ref]
    o_ref = torch.optim.Adam(ref, lr=3e-3, betas=(0.9, 0.99), eps=1e-8, weight_decay=1e-2)
    o_mine = mtt_amd.optim.FusedClipAdam(mine, lr=3e-3, betas=(0.9, 0.99), eps=1e-8, weight_decay=1e-2, max_norm=2.0)
    s_ref = mtt_amd.optim.PolynomialLR(o_ref, max_iterations=10, gamma=0.9)
    s_mine = mtt_amd.optim.PolynomialLR(o_mine, max_iterations=10, gamma=0.9)
    for it in range(3):
        gs = [torch.randn(s) * (3.0 if it == 0 else 0.05) for s in shapes]      # first step clips, later ones do not
        for q, r, g in zip(mine, ref, gs):
            q.grad, r.grad = g.clone(), g.clone()
        n_ref = torch.nn.utils.clip_grad_norm_(ref, 2.0)
        o_ref.step()
        n_mine = o_mine.step()
        s_ref.step(), s_mine.step()
        assert abs(float(n_ref) - float(n_mine)) < 1e-4 * float(n_ref)
        for q, r in zip(mine, ref):
            assert float((q.detach() - r.detach()).abs().max()) < 2e-6, it
    sd = o_mine.state_dict()
    assert set(sd["state"][0]) == {"step", "exp_avg", "exp_avg_sq"}
    o_ref.load_state_dict(sd)                                                     # layouts are interchangeable


def test_fused_clip_adam_capturable_reads_step_scalars_from_memory(emulated, monkeypatch):
    """capturable=True (mtt_adam_desc.hyper: the bias corrections / learning rate come from a memory pair) takes the same steps as the
    launch-constant form, with a learning-rate schedule in between."""
    import mtt_amd
    monkeypatch.setattr(mtt_amd.ops, "adam_chunk", lambda: 65536)
    torch.manual_seed(1)
    shapes = [(3000,), (33, 17)]
    a = [torch.nn.Parameter(torch.randn(s)) for s in shapes]
    b = [torch.nn.Parameter(q.detach().clone()) for q in a]
    oa = mtt_amd.optim.FusedClipAdam(a, lr=3e-3, weight_decay=1e-2, max_norm=2.0)
    ob = mtt_amd.optim.FusedClipAdam(b, lr=3e-3, weight_decay=1e-2, max_norm=2.0, capturable=True)
    sa, sb = mtt_amd.optim.PolynomialLR(oa, max_iterations=10), mtt_amd.optim.PolynomialLR(ob, max_iterations=10)
    for it in range(3):
        for q, r in zip(a, b):
            g = torch.randn(q.shape)
            q.grad, r.grad = g.clone(), g.clone()
        oa.step(), ob.step()
        sa.step(), sb.step()
        for q, r in zip(a, b):
            assert torch.equal(q.detach(), r.detach()), it
    with pytest.raises(RuntimeError):
        ob.prepare_replay()                                  # nothing was captured


def test_weight_gradient_slice_count_fills_one_round_of_cus():
    """autograd_path._tn_splits: 256 x 256 weight-gradient tiles occupy a whole CU, so the slice count must not spill a few workgroups into a
    second round (48 tiles: 5 slices = 240 workgroups, not ceil(256 / 48) = 6 -> 288), and few-tile outputs still get many slices."""
    import mtt_amd
    from mtt_amd import autograd_path
    ts = autograd_path._tn_splits
    rows = 63 * 1030
    assert ts(48, rows) == 5 and ts(64, rows) == 4 and ts(16, rows) == 16
    for tiles in (1, 3, 9, 12, 36, 48, 64, 100, 150, 191):
        S = ts(tiles, rows)
        assert 2 <= S <= 32
        rounds = -(-tiles * S // 256)
        assert tiles * S > 256 * (rounds - 1) + 0.5 * 256 or rounds == 1, (tiles, S)      # the last round is at least half full
    assert ts(4, 8240) < ts(4, rows)                                                        # short reductions take fewer slabs


def test_load_state_dict_keeps_the_moment_tensors_a_captured_step_updates(emulated, monkeypatch):
    """After a capture the graph holds the addresses of exp_avg / exp_avg_sq: load_state_dict must fill THOSE tensors instead of replacing them."""
    import mtt_amd
    monkeypatch.setattr(mtt_amd.ops, "adam_chunk", lambda: 65536)
    torch.manual_seed(2)
    a = [torch.nn.Parameter(torch.randn(50)), torch.nn.Parameter(torch.randn(7, 3))]
    b = [torch.nn.Parameter(q.detach().clone()) for q in a]
    oa = mtt_amd.optim.FusedClipAdam(a, lr=1e-2, capturable=True)
    ob = mtt_amd.optim.FusedClipAdam(b, lr=1e-2, capturable=True)
    for q, r in zip(a, b):
        q.grad = torch.randn(q.shape)
        r.grad = torch.randn(r.shape)
    oa.step(), ob.step()
    ob._captured = [(0, 0, list(b))]                       # what step() records under stream capture
    held = [(ob.state[r]["exp_avg"], ob.state[r]["exp_avg_sq"]) for r in b]
    ob.load_state_dict(oa.state_dict())
    for q, r, (m, v) in zip(a, b, held):
        assert ob.state[r]["exp_avg"] is m and ob.state[r]["exp_avg_sq"] is v
        assert torch.equal(m, oa.state[q]["exp_avg"]) and torch.equal(v, oa.state[q]["exp_avg_sq"])
        assert float(ob.state[r]["step"]) == float(oa.state[q]["step"])


def test_graphed_train_step_refuses_what_it_cannot_record():
    import mtt_amd
    w = torch.nn.Parameter(torch.zeros(4))
    x = torch.zeros(1, 3, 8, 8)
    with pytest.raises(ValueError, match="capturable"):
        mtt_amd.graphs.GraphedTrainStep(None, None, mtt_amd.optim.FusedClipAdam([w]), x, {})
    with pytest.raises(ValueError, match="GPU"):
        mtt_amd.graphs.GraphedTrainStep(None, None, mtt_amd.optim.FusedClipAdam([w], capturable=True), x, {})


def _loss_case(device, B=2, H=12, W=10):
    import mtt_amd
    p = mtt_amd.factory.make_p(mtt_amd.factory.TASK_ORDER, (H, W))
    g = torch.Generator().manual_seed(4)
    pred = {t: (torch.randn(B, p.TASKS.NUM_OUTPUT[t], H, W, generator=g) * 2).to(device).requires_grad_(True) for t in p.TASKS.NAMES}
    gt = mtt_amd.losses.synthetic_targets(p, B, H, W, device, seed=5)
    return p, pred, gt


def check_fused_losses(device, tol):
    import mtt_amd
    p, pred, gt = _loss_case(device)
    ref_pred = {t: v.detach().cpu().clone().requires_grad_(True) for t, v in pred.items()}
    ref = losses_oracle.MultiTaskLoss(p, p.TASKS.NAMES)(ref_pred, {t: v.cpu() for t, v in gt.items()})
    out = mtt_amd.losses.FusedMultiTaskLoss(p, p.TASKS.NAMES)(pred, gt)
    ref["total"].backward()
    out["total"].backward()
    for t in p.TASKS.NAMES + ["total"]:
        a, b = float(out[t].detach()), float(ref[t].detach())
        assert abs(a - b) <= tol * max(1.0, abs(b)), (t, a, b)
    for t in p.TASKS.NAMES:
        a, b = pred[t].grad.cpu(), ref_pred[t].grad
        assert float((a - b).norm()) <= tol * 10 * float(b.norm()) + 1e-12, t


def test_fused_losses_match_the_restated_criterion_on_emulator(emulated):
    """FusedMultiTaskLoss (mtt_loss_* through the emulator) vs the torch restatement of the reference criterion: values and
    gradients w.r.t. the logits for all six PASCAL task losses (ignore labels, class-frequency weights, pos_weight, normalisation)."""
    check_fused_losses("cpu", 1e-6)


def test_flax_vit_checkpoint_import_matches_reference_loader():
    """checkpoints.load_flax_vit_npz / filter_state_dict vs tensors produced by the unmodified reference loaders
    (tests/golden/make_ckpt_golden.py): layout rules, q/k/v packing, bicubic position-embedding resize, class token handling."""
    import numpy as np
    import mtt_amd
    from tests.golden.make_ckpt_golden import STRIDE, fake_flax_vit
    gold = np.load(os.path.join(os.path.dirname(__file__), "golden", "ckpt_import.npz"))
    for kind, name, seed in (("TP", "mini_ctr", 1), ("IP", "mini", 2)):
        cfg = dict(configs.taskprompter(name) if kind == "TP" else configs.invpt(name), backbone="nano")
        model = conftest.build_product_model(cfg, "x3")
        C, depth, heads, _ = configs.VIT["nano"]
        loaded = mtt_amd.checkpoints.load_flax_vit_npz(model.backbone, fake_flax_vit(C, depth, heads, 3, seed=seed))
        sd = model.backbone.state_dict()
        expect = [k[len(kind) + 8:] for k in gold.files if k.startswith(f"{kind}/expect/")]
        assert sorted(expect) == sorted(loaded)
        for k in expect:
            got = sd[k].numpy() if k == "pos_embed" else sd[k].numpy().reshape(-1)[::STRIDE]
            assert np.abs(got - gold[f"{kind}/expect/{k}"]).max() < 1e-5, (kind, k)
        if kind == "TP":
            raw = {"model": {k[len("TP/filter_in/"):]: torch.from_numpy(gold[k]) for k in gold.files if k.startswith("TP/filter_in/")}}
            filt = mtt_amd.checkpoints.filter_state_dict(raw, model.backbone)
            for k, v in filt.items():
                assert np.abs(v.numpy() - gold[f"TP/filter_out/{k}"]).max() < 1e-5, k


def test_training_trajectory_harness_on_emulator(emulated):
    """The harness of tests/test_gpu_train.py::test_mixed_precision_training_trajectory_... (200 steps on the device) run for 8 steps on
    the emulator: x3f and x3 against the oracle's fp32 training from the same state, the reference's criterion, clip + Adam."""
    import train_check
    curves = train_check.trajectory_check("mini_ctr", ["x3f", "x3"], "cpu", steps=8, lr=2e-4)
    gaps = train_check.trajectory_gaps(curves, window=4)
    assert curves["x3"][0] == pytest.approx(curves["oracle"][0], rel=1e-5) and curves["x3f"][0] == pytest.approx(curves["oracle"][0], rel=1e-5)
    assert gaps["x3"][0] < 2e-3 and gaps["x3f"][0] < 5e-3, gaps
    assert curves["oracle"][-1] < curves["oracle"][0]


def test_multi_step_training_rebuilds_weight_packs(emulated):
    """Three optimizer steps: FusedClipAdam writes the parameters through raw pointers, so every cached weight pack (bf16 / padded /
    transposed / BN-folded) must be invalidated — the product's losses and final parameters track clip_grad_norm_ + torch.optim.Adam
    on the oracle.  (A stale pack cache makes every step after the first reuse step-0 weights: the loss would not move.)"""
    import train_check
    losses, worst = train_check.multi_step_check("mini_ctr", "x3", "cpu", steps=3)
    for lp, lo in losses:
        assert abs(lp - lo) <= 2e-4 * max(1.0, abs(lo)), losses
    assert abs(losses[0][0] - losses[2][0]) > 1e-3, losses          # the objective actually moves
    assert worst < 2e-3, worst


def test_pack_cache_sees_raw_pointer_updates(emulated):
    import mtt_amd
    w = torch.nn.Parameter(torch.randn(16, 24))
    prec = mtt_amd.ops.Prec("bf16")
    a = mtt_amd.ops.pack_linear([w], prec, "t").clone()
    opt = mtt_amd.optim.FusedClipAdam([w], lr=0.1)
    w.grad = torch.ones_like(w)
    v0 = w._version
    opt.step()
    assert w._version > v0
    b = mtt_amd.ops.pack_linear([w], prec, "t")
    assert float((a.float() - b.float()).abs().max()) > 0.05
    # load_state_dict replaces the moment tensors: the pointer tables must follow them
    sd = opt.state_dict()
    opt2 = mtt_amd.optim.FusedClipAdam([w], lr=0.1)
    w.grad = torch.ones_like(w)
    opt2.step()
    opt2.load_state_dict(sd)
    m_before = opt2.state[w]["exp_avg"].clone()
    opt2.step()
    assert not torch.equal(opt2.state[w]["exp_avg"], m_before)


def test_heads_standalone_and_mixed_head_sets(emulated):
    """A head module is callable on its own (`head(x)`, reference layout in / out) and the wrapper accepts any mix of ConvHead,
    DEConvHead and foreign modules (taskprompter_wrapper.py:29-38), each task through its own head."""
    import mtt_amd
    from oracle import taskprompter_oracle as tpo
    tp = mtt_amd.taskprompter
    cfg = configs.taskprompter("mini_deconv")
    meta, _ = conftest.load_golden("mini_deconv")
    sd = weights.synth_state_dict(meta["contract"], 0)
    model = conftest.build_product_model(cfg, "x3")
    model.load_state_dict(sd, strict=True)
    model.eval()
    x = weights.synth_images(1, cfg["img_size"], 3)
    with torch.no_grad():
        full = model(x)
        feats, _ = model.backbone(x)
        for t, n in cfg["tasks"]:
            y = model.heads[t](feats[t])                               # DEConvHead alone: [B, n, 8h, 8w]
            ref = tpo.head_forward(sd, f"heads.{t}", "deconv", feats[t].float())
            assert y.shape == ref.shape and float((y - ref).norm() / ref.norm()) < 5e-5, t
            up = torch.nn.functional.interpolate(y, size=tuple(x.shape[-2:]), mode="bilinear", align_corners=False)
            assert float((up - full[t]).norm() / full[t].norm()) < 5e-5, t
    # mixed: semseg keeps its DEConvHead, depth gets a ConvHead, plus a foreign torch head for a third pseudo-task
    F = cfg["final_embed_dim"]
    p2 = model.backbone.p
    heads = torch.nn.ModuleDict({"semseg": model.heads["semseg"], "depth": tp.ConvHead(F, 1)})
    mixed = tp.TaskPrompterWrapper(p2, model.backbone, heads).eval()
    with torch.no_grad():
        out = mixed(x)
        assert float((out["semseg"] - full["semseg"]).abs().max()) < 1e-5
        hd = heads["depth"]
        sdh = {"h." + k: v for k, v in hd.state_dict().items()}
        ref = tpo.head_forward(sdh, "h", "conv", feats["depth"].float())
        ref = torch.nn.functional.interpolate(ref, size=tuple(x.shape[-2:]), mode="bilinear", align_corners=False)
        assert float((out["depth"] - ref).norm() / ref.norm()) < 5e-5
    # a training step through a standalone head reaches its parameters
    hd.train()
    y = hd(feats["depth"].float().requires_grad_(True))
    y.square().mean().backward()
    assert all(q.grad is not None for q in hd.parameters())


def test_fused_upsample_conv_matches_the_materialised_path(emulated):
    """ConvHeads fuse the backbone's x4 bilinear resize into their 3x3 conv ("taps first": GEMM with the nine stacked tap matrices on the
    h x w map + mtt_upconv4_expand, backward mtt_upconv4_gather).  It must give what resize-then-conv gives — outputs (eval, BN folded
    into the expansion's scale / bias), train-mode outputs and every parameter gradient — and never build the upsampled stack."""
    import mtt_amd
    import train_check
    tp = mtt_amd.taskprompter
    cfg = configs.taskprompter("mini_ctr")
    meta, _ = conftest.load_golden("mini_ctr")
    sd = weights.synth_state_dict(meta["contract"], 0)
    x = weights.synth_images(2, cfg["img_size"], 5)
    names = []
    real = mtt_amd.ops.call
    mtt_amd.ops.call = lambda name, **kw: (names.append(name), real(name, **kw))[1]
    res = {}
    try:
        for fuse in (True, False):
            tp.TaskPrompterWrapper.fuse_upsample = fuse
            del names[:]
            model = conftest.build_product_model(cfg, "x3")
            model.load_state_dict(sd, strict=True)
            model.eval()
            with torch.no_grad():
                ev = model(x)
            n_up = sum(1 for n in names if n == "upconv4_expand")
            assert n_up == (1 if fuse else 0)
            model.train()
            out = model(x)
            sum((v * torch.linspace(-1, 1, v.numel()).view(v.shape)).sum() for v in out.values()).backward()
            assert ("upconv4_gather" in names) == fuse
            res[fuse] = (ev, {k: v.detach() for k, v in out.items()}, {k: q.grad.clone() for k, q in model.named_parameters() if q.grad is not None})
    finally:
        tp.TaskPrompterWrapper.fuse_upsample = True
        mtt_amd.ops.call = real
    for part in (0, 1, 2):
        a, b = res[True][part], res[False][part]
        assert a.keys() == b.keys()
        floor = 1e-3 * max(float(v.norm()) for v in b.values())       # a bias in front of a BatchNorm has a zero gradient: rounding noise only
        for k in a:
            den = float(b[k].norm())
            if den > floor:
                assert float((a[k] - b[k]).norm()) / den < 2e-4, (part, k)


# ---- TaskPrompter-Swin (forward path) -------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["mini_swin", "mini_swin_pad"])
@pytest.mark.parametrize("prec,tol", [("x3", 1e-4), ("bf16", 5e-2)])
def test_swin_wiring_matches_reference_golden(emulated, name, prec, tol):
    """The product's TaskPrompterSwin schedule (window tables, gathers, window / channel attention descriptors, patch merging, commuted
    1x1 convs, multi-scale accumulation, heads) on the ABI emulator against the UNMODIFIED reference's outputs; the state-dict contract
    (names, shapes, order, geometry-derived buffers) must be the reference's."""
    from oracle import swin_oracle as swo
    cfg = configs.swin(name)
    meta, gold = conftest.load_golden(name)
    model = conftest.build_product_model(cfg, prec)
    assert [(k, list(v.shape)) for k, v in model.state_dict().items()] == [(k, list(s)) for k, s in meta["contract"]]
    res = model.load_state_dict(weights.synth_state_dict(meta["contract"], 0), strict=False)
    assert not res.unexpected_keys and all(k.rsplit(".", 1)[-1] in weights.DERIVED_BUFFERS for k in res.missing_keys)
    for k, v in model.state_dict().items():                     # the derived buffers are what the (pinned) oracle computes
        if k.endswith("relative_position_index"):
            assert torch.equal(v, swo.relative_position_index(math.isqrt(v.shape[0])))
    model.eval()
    with torch.no_grad():
        out = model(weights.synth_images(meta["batch"], cfg["img_size"], 1))
    es = meta["eval_stride"]
    for t, n in cfg["tasks"]:
        g = torch.from_numpy(gold[f"eval/{t}"])
        assert out[t].dtype == torch.float32 and tuple(out[t].shape[-2:]) == tuple(cfg["img_size"])
        e = float((out[t][:, :, ::es, ::es] - g).norm() / g.norm())
        assert e < tol, (t, prec, e)


@pytest.mark.parametrize("name", ["mini_swin", "mini_swin_pad"])
def test_swin_training_gradients_on_emulator(emulated, name):
    """swin_autograd.py (window gather adjoints, window-attention backward incl. the relative-position-bias table and the raw-logit
    gradient, channel attention, patch merging of features / maps / prompts, 32- and 64-channel head modulation) on the ABI emulator
    against the oracle's autograd: every parameter gradient within 1e-3 relative, train-mode forward within 5e-5."""
    import train_check
    fwd, errs, dead = train_check.swin_grad_errors(name, "x3", "cpu")
    assert max(fwd.values()) < 5e-5, fwd
    worst, med = train_check.summarize(errs)
    assert worst[0] < 1e-3 and med < 1e-4, (worst, med)
    assert not dead, dead[:5]


def check_swin_x3f_split_planes(name, device, monkeypatch):
    """(shared with tests/test_gpu_train.py) TaskPrompter-Swin in the x3f mode (round 6): the stage Linears (one split pass + the split-plane
    GEMM), the task features (modulate and the fea_decode epilogue write hi / lo planes) and the matrix-core window attention
    (mtt_winattn_desc.mfma) against the oracle — eval forward fp32-class, training gradients within the per-parameter bound of the bf16
    backward; with the row thresholds lowered so that the miniature takes the paths Swin-B takes."""
    import mtt_amd
    import train_check
    from oracle import swin_oracle as swo
    monkeypatch.setattr(mtt_amd.autograd_path, "AUTO_SPLIT_MIN_ROWS", 64)
    monkeypatch.setattr(mtt_amd.taskprompter_swin.TaskPrompterSwin, "SPLIT_MIN_ROWS", 64)
    seen = []
    inner = mtt_amd.ops.call
    monkeypatch.setattr(mtt_amd.ops, "call", lambda n, **kw: (seen.append((n, kw.get("a_dtype"), kw.get("mfma"), kw.get("out_lo") is not None)), inner(n, **kw))[1])
    cfg = configs.swin(name)
    model = conftest.build_product_model(cfg, "x3f", device)
    contract = [(k, list(v.shape)) for k, v in model.state_dict().items() if k.rsplit(".", 1)[-1] not in weights.DERIVED_BUFFERS]
    sd = weights.synth_state_dict(contract, 0)
    model.load_state_dict({k: v.to(device) for k, v in sd.items()}, strict=False)
    model.eval()
    x = weights.synth_images(2, cfg["img_size"], 1)
    with torch.no_grad():
        out = model(x.to(device))
    ref = swo.forward(sd, cfg, x)
    for t, _ in cfg["tasks"]:
        e = float((out[t].cpu() - ref[t]).norm() / ref[t].norm())
        assert e < 5e-5, (t, e)
    n_split = sum(1 for n, adt, _, _ in seen if n == "gemm" and adt == 2)
    assert all(m == 1 for n, _, m, _ in seen if n == "winattn_fwd")
    if name == "mini_swin_sp":
        assert n_split >= 6 * 4 + 4 * 2, n_split           # qkv / proj / fc1 / fc2 of the six blocks with >= 64 channels, two task-feature GEMMs per level
        assert sum(1 for n, _, _, lo in seen if n == "modulate" and lo) == 4
    del seen[:]
    fwd, errs, dead = train_check.swin_grad_errors(name, "x3f", device, contract=contract)
    assert max(fwd.values()) < 5e-5, fwd
    worst, med = train_check.summarize(errs, floor=1e-4)
    print(f"PARITY swin-train {name} x3f fwd {max(fwd.values()):.3e} grad median {med:.3e} worst {worst[0]:.3e} ({worst[1]})")
    assert med < 3e-2, (worst, med)
    train_check.assert_per_param(errs, "x3f")
    assert all(m == 1 for n, _, m, _ in seen if n in ("winattn_fwd", "winattn_bwd")) and any(n == "winattn_bwd" for n, *_ in seen)


@pytest.mark.parametrize("name", ["mini_swin_sp", "mini_swin"])
def test_swin_x3f_split_planes_on_emulator(emulated, monkeypatch, name):
    check_swin_x3f_split_planes(name, "cpu", monkeypatch)


@pytest.mark.parametrize("family,name,prec", [("taskprompter", "mini_ctr", "x3f"), ("taskprompter", "mini_deconv", "bf16"), ("invpt", "mini8", "x3f"),
                                              ("swin", "mini_swin", "x3f"), ("swin", "mini_swin_pad", "bf16")])
def test_training_steps_do_not_accumulate_tensors(emulated, family, name, prec):
    """Five training iterations: the bytes held by live tensors must be constant from the second step on.  (Until round 6 WinAttnFn kept its OUTPUT on
    ctx — a reference cycle node -> ctx -> out -> grad_fn that no collector breaks: a block's qkv + out leaked per step, GBs at the Swin-B shape.)"""
    import gc
    import mtt_amd
    cfg = getattr(configs, family)(name)
    model = conftest.build_product_model(cfg, prec)
    model.train()
    opt = torch.optim.SGD(model.parameters(), lr=1e-4)
    x = weights.synth_images(2, cfg["img_size"], 1)
    held = []
    for _ in range(5):
        out = model(x)
        loss = sum(v.float().sum() for v in out.values() if torch.is_tensor(v)) + sum(v.float().sum() for v in (out.get("inter_preds") or {}).values())
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        mtt_amd.ops.bump_param_epoch()
        del out, loss
        gc.collect()
        held.append(sum(o.numel() * o.element_size() for o in gc.get_objects() if torch.is_tensor(o)))
    assert held[1] == held[2] == held[3] == held[4], held


def test_swin_training_with_droppath_masks(emulated):
    """The block's four independent DropPath draws (pixels / prompts x attention / MLP), injected into product and oracle alike."""
    import train_check
    cfg = configs.swin("mini_swin")
    drop = train_check.swin_drop_masks(cfg, 2)
    fwd, errs, dead = train_check.swin_grad_errors("mini_swin", "x3", "cpu", drop=drop)
    assert max(fwd.values()) < 5e-5, fwd
    worst, med = train_check.summarize(errs)
    assert worst[0] < 1e-3 and med < 1e-4, (worst, med)


def test_swin_window_tables_are_consistent():
    """part / rev / pix are mutually inverse views of one permutation-with-padding, and equal roll + pad + window_partition."""
    import mtt_amd
    sw = mtt_amd.taskprompter_swin
    for res, window, shifted, T in (((7, 9), 5, True, 3), ((8, 12), 4, True, 2), ((4, 6), 4, True, 2), ((12, 24), 12, False, 1)):
        ws, shift, Hp, Wp = sw.block_geometry(res, window, shifted)
        part, pix, rev = sw.window_tables(res, ws, shift, Hp, Wp, T, "cpu")
        H, W = res
        Nw = T + ws * ws
        assert torch.equal(part.view(-1, Nw)[:, T:][pix >= 0].long() - T, pix[pix >= 0].long())
        assert torch.equal(part[rev.long()].long(), T + torch.arange(H * W))                 # every pixel sits where rev says
        x = torch.arange(H * W, dtype=torch.float32).view(1, H, W, 1) + 1
        xp = torch.nn.functional.pad(x, (0, 0, 0, Wp - W, 0, Hp - H))
        xs = torch.roll(xp, (-shift, -shift), (1, 2)) if shift else xp
        win = xs.view(1, Hp // ws, ws, Wp // ws, ws, 1).permute(0, 1, 3, 2, 4, 5).reshape(-1, ws * ws)
        assert torch.equal(torch.where(pix >= 0, pix + 1, torch.zeros_like(pix)).float(), win)


def test_persistent_packs_match_torch_relayouts_on_emulator(emulated):
    """ops.seg_pack / mtt_segcopy (emulated): every pack layout bit-exact against a plain torch re-layout; one refresh per parameter update."""
    import pack_check
    pack_check.check_packs("cpu")
    pack_check.check_refresh("cpu")
    pack_check.check_unpack("cpu")


def test_bench_pmc_traffic_record_is_refused_when_the_kernel_source_changed(tmp_path, monkeypatch):
    """bench.py reports roofline.traffic from the committed PMC passes only while csrc/gemm.hip is the file they were measured on."""
    import hashlib
    import json as _json
    import bench
    (tmp_path / "profiles").mkdir()
    src = tmp_path / "multi-task-transformer_amd" / "csrc"
    src.mkdir(parents=True)
    (src / "gemm.hip").write_text("// kernel source, version A\n")
    sha = hashlib.sha256((src / "gemm.hip").read_bytes()).hexdigest()[:16]
    rec = {"gemm_ring3_kernel": dict(hbm_bytes_per_launch=123, commit="abc1234", csrc_sha=sha, source="pmc")}
    (tmp_path / "profiles" / "pmc_traffic.json").write_text(_json.dumps(rec))
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    got = bench._pmc_traffic("gemm_ring3_kernel")
    assert got["hbm_bytes_per_launch"] == 123 and got["commit"] == "abc1234"
    assert bench._pmc_traffic("gemm_dma_kernel<1>").get("hbm_bytes_per_launch") is None        # no record for that kernel
    (src / "gemm.hip").write_text("// kernel source, version B\n")
    stale = bench._pmc_traffic("gemm_ring3_kernel")
    assert stale.get("hbm_bytes_per_launch") is None and "STALE" in stale["note"] and stale["commit"] == "abc1234"


def test_pmc_record_was_measured_on_this_trees_gemm_source():
    """the committed PMC traffic record (profiles/pmc_traffic.json, the source of the bench line's roofline.traffic) must have
    been measured on the gemm.hip of THIS tree — the re-measurement is the last GPU step of a round that touched the file.  A stale record
    is caught here instead of surfacing as `traffic: null` in the driver's line."""
    import hashlib
    import json as _json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sha = hashlib.sha256(open(os.path.join(root, "multi-task-transformer_amd", "csrc", "gemm.hip"), "rb").read()).hexdigest()[:16]
    rec = _json.load(open(os.path.join(root, "profiles", "pmc_traffic.json")))
    for kernel in ("gemm_ring3_kernel", "gemm_dma_kernel<1>"):
        assert rec[kernel]["csrc_sha"] == sha, (kernel, rec[kernel]["csrc_sha"], sha)
        assert rec[kernel]["hbm_bytes_per_launch"] > 0 and rec[kernel]["launches"] > 0
    import bench
    assert bench._pmc_traffic("gemm_ring3_kernel")["hbm_bytes_per_launch"] == rec["gemm_ring3_kernel"]["hbm_bytes_per_launch"]
    ident = bench._source_id()
    assert ident["head"] and ident["tree_sha"] == bench.source_tree_sha()


def test_gemm_policy_routes_the_round5_kernels():
    """mtt_gemm_variant (the library's dispatch policy, a pure function of the descriptor; no GPU needed): the bf16 3x3 convs of the
    benchmark's shapes go to the LDS-DMA ring kernel (12), the split-plane ones to gemm_ring3_kernel<true> (9); channel pitches that are
    not multiples of 32, small maps and forced-general calls stay on the register-staged kernel (0)."""
    import mtt_amd
    gv = mtt_amd._lib.gemm_variant
    def conv(M, N, Cp, **kw):
        d = dict(M=M, N=N, K=9 * Cp, a_op=2, b_op=0, a_dtype=1, b_dtype=1, d_dtype=1, prec=0, lda=Cp, ldb=9 * Cp, ldd=(N + 7) // 8 * 8, batch=6,
                 conv=dict(H=32, W=32, C=Cp - 2, Cp=Cp, dil=1, flip=1))
        d.update(kw)
        return gv(**d)
    assert conv(63 * 1024, 350, 352) == 12                     # fea_fuse 3x3 input gradient (x3f / bf16 backward)
    assert conv(32 * 16384, 576, 576, conv=dict(H=128, W=128, C=576, Cp=576, dil=1, flip=0)) == 12      # InvPT mt_proj, bf16 forward
    assert conv(63 * 1024, 300, 304) == 0                      # pitch 304: a 32-deep K step would straddle two taps
    assert conv(1024, 350, 352) == 0                           # small map: 128 x 128 tiles fill the chip better
    assert conv(63 * 1024, 350, 352, variant=1) == 0           # MTT_GEMM_GENERAL
    assert conv(1024, 40, 32, variant=3) == 12                 # forced (the op tests' small shapes)
    assert conv(63 * 1024, 350, 352, a_dtype=2, b_dtype=2, prec=1, A_lo=torch.zeros(1), B_lo=torch.zeros(1), d_dtype=0) == 9
    # the prediction dgrad of TaskHeadsFn: K = pad8(n) on the 128-row LDS-DMA kernel when forced, the general kernel otherwise
    head = dict(M=63 * 16384, N=352, K=24, a_op=0, b_op=0, a_dtype=1, b_dtype=1, d_dtype=1, prec=0, lda=24, ldb=24, ldd=352, batch=1)
    assert gv(**head, variant=4) == 4 and gv(**head) in (0, 4)


def test_multi_scale_sum_node_equals_the_separate_resizes(emulated):
    """invpt_autograd.MultiScaleSumFn (the stage outputs resized and summed in ONE node: in-kernel accumulation, identity scale handed
    through) against BilinearFn per stage + adds: same values, same input gradients."""
    import importlib
    ia = importlib.import_module("multi-task-transformer_amd.invpt_autograd")
    ap = importlib.import_module("multi-task-transformer_amd.autograd_path")
    g = torch.Generator().manual_seed(3)
    T, B, C, th, tw = 2, 2, 16, 8, 12
    sizes = ((2, 3), (4, 6), (8, 12))
    ys = [torch.randn(T, B * h * w, C, generator=g).requires_grad_(True) for h, w in sizes]
    ys2 = [y.detach().clone().requires_grad_(True) for y in ys]
    acc = ia.MultiScaleSumFn.apply((B, C, th, tw, sizes), *ys)
    ref = None
    for y, (h, w) in zip(ys2, sizes):
        r = ap.BilinearFn.apply(y, (B, C, h, w, th, tw), torch.float32, False)
        ref = r if ref is None else ref + r
    assert torch.allclose(acc, ref, rtol=1e-6, atol=1e-6)
    wgt = torch.randn(acc.shape, generator=g)
    (acc * wgt).sum().backward()
    (ref * wgt).sum().backward()
    for a, b in zip(ys, ys2):
        assert torch.allclose(a.grad, b.grad, rtol=1e-5, atol=1e-6)


def test_pretrained_true_reads_timms_cache_and_never_downloads(tmp_path, monkeypatch):
    """`pretrained=True` is what the reference's get_backbone passes (TaskPrompter/utils/common_config.py:22): the constructors look for
    the variant's `.npz` where timm 0.5.4's download_cached_file keeps it (<torch hub dir>/checkpoints/<file name of the URL>) and fail with
    that path when it is not there — no network access, no silent random initialisation."""
    import mtt_amd
    monkeypatch.setenv("TORCH_HOME", str(tmp_path))
    path = mtt_amd.checkpoints.cached_pretrained_path("vit_base_patch16_384")
    assert path.startswith(str(tmp_path)) and path.endswith("res_384.npz") and "B_16" in os.path.basename(path)
    p = mtt_amd.factory.make_p(["semseg", "depth"], (64, 64), backbone="TaskPrompter_vitB", head="conv", embed_dim=24, final_embed_dim=24,
                               num_output=dict(semseg=5))
    with pytest.raises(RuntimeError) as e:
        mtt_amd.taskprompter.taskprompter_vit_base_patch16_384(p=p, pretrained=True, drop_path_rate=0.0, img_size=(64, 64))
    assert path in str(e.value) and "never downloads" in str(e.value)
    with pytest.raises(RuntimeError):
        mtt_amd.checkpoints.load_cached_pretrained(object(), "no_such_variant")
