"""-m gpu: full TaskPrompter forward through the C ABI on the MI355X against the golden outputs of the
unmodified reference (tests/golden/*.npz) and the CPU oracle.

Tolerance (BASELINE.json north_star): 1e-3 relative (norm-wise, per task head) in fp32-class arithmetic —
checked in MTT_PREC_X3 mode.  The bf16 throughput mode is checked against a measured-error bound of 4e-2
(bf16 operands carry 2^-9 relative rounding per GEMM; after 4-24 blocks ~1e-2)."""
import pytest
import torch

import conftest
from oracle import configs, taskprompter_oracle as tpo, weights

REL_TOL_X3 = 1e-3
REL_TOL_BF16 = 4e-2


def _rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["mini_ctr", "mini_win", "mini_deconv"])
@pytest.mark.parametrize("prec,tol", [("x3", REL_TOL_X3), ("x3f", REL_TOL_X3), ("bf16", REL_TOL_BF16)])
def test_forward_matches_reference_golden(name, prec, tol):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    cfg = configs.taskprompter(name)
    meta, gold = conftest.load_golden(name)
    model = conftest.build_product_model(cfg, prec, "cuda")
    model.load_state_dict(weights.synth_state_dict(meta["contract"], 0), strict=True)
    model.eval()
    x = weights.synth_images(meta["batch"], cfg["img_size"], 1).cuda()
    with torch.no_grad():
        out = model(x)
    for t, n in cfg["tasks"]:
        g = torch.from_numpy(gold[f"eval/{t}"])
        assert out[t].shape == g.shape
        assert _rel(out[t].cpu(), g) < tol, (t, _rel(out[t].cpu(), g))
    # train-mode forward: batch-statistic BatchNorm + running-stat update
    model.train()
    x2 = weights.synth_images(2, cfg["img_size"], 2).cuda()
    with torch.no_grad():
        out = model(x2)
    for t, n in cfg["tasks"]:
        g = torch.from_numpy(gold[f"train/{t}"])
        assert _rel(out[t].cpu()[:, :, ::2, ::2], g) < tol, t
    if prec == "x3":
        sd = model.state_dict()
        for k in sd:
            if k.endswith("running_mean") or k.endswith("running_var"):
                g = torch.from_numpy(gold[f"bn/{k}"])
                assert float((sd[k].cpu() - g).abs().max()) < 1e-3 * max(1.0, float(g.abs().max())), k


@pytest.mark.gpu
def test_forward_matches_oracle_on_fresh_inputs_larger_batch():
    """oracle (CPU restatement) vs HIP path on inputs that are not in the fixtures; B = 3, ragged tiles."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    cfg = configs.taskprompter("mini_ctr")
    meta, _ = conftest.load_golden("mini_ctr")
    sd = weights.synth_state_dict(meta["contract"], 5)
    model = conftest.build_product_model(cfg, "x3", "cuda")
    model.load_state_dict(sd, strict=True)
    model.eval()
    x = weights.synth_images(3, cfg["img_size"], 9)
    with torch.no_grad():
        ref = tpo.forward(sd, cfg, x)
        out = model(x.cuda())
    for t, _ in cfg["tasks"]:
        assert _rel(out[t].cpu(), ref[t]) < REL_TOL_X3, t


@pytest.mark.gpu
@pytest.mark.parametrize("prec,tol", [("x3", REL_TOL_X3), ("x3f", REL_TOL_X3), ("bf16", REL_TOL_BF16)])
def test_invpt_forward_matches_reference_golden(prec, tol):
    """InvPT (ViT taps + TransformerDecoder + InvPT stages + MLPHead): eval and train-mode forward vs the unmodified reference."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from test_host_cpu import _invpt_outputs_vs_golden
    cfg = configs.invpt("mini")
    meta, gold = conftest.load_golden("mini")
    model = conftest.build_product_model(cfg, prec, "cuda")
    model.load_state_dict(weights.synth_state_dict(meta["contract"], 0), strict=True)
    _invpt_outputs_vs_golden(model, cfg, meta, gold, tol, device="cuda")


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["mini_swin", "mini_swin_pad"])
@pytest.mark.parametrize("prec,tol", [("x3", REL_TOL_X3), ("bf16", 5e-2)])
def test_swin_forward_matches_reference_golden(name, prec, tol):
    """TaskPrompter-Swin forward (shifted / padded windows with prompts, channel attention, patch merging, multi-scale fusion, Conv and
    DEConv heads) through the HIP kernels against the unmodified reference's outputs."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    cfg = configs.swin(name)
    meta, gold = conftest.load_golden(name)
    model = conftest.build_product_model(cfg, prec, "cuda")
    model.load_state_dict(weights.synth_state_dict(meta["contract"], 0), strict=False)
    model.eval()
    x = weights.synth_images(meta["batch"], cfg["img_size"], 1).cuda()
    with torch.no_grad():
        out = model(x)
    es = meta["eval_stride"]
    for t, n in cfg["tasks"]:
        g = torch.from_numpy(gold[f"eval/{t}"])
        e = _rel(out[t].cpu()[:, :, ::es, ::es], g)
        print(f"PARITY swin {name} {prec} {t} {e:.3e}")
        assert e < tol, (t, e)
