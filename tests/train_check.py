"""Shared helper: one training step's forward + backward of the product vs the oracle's autograd."""
import collections

import torch

import conftest
from oracle import configs, losses_oracle, taskprompter_oracle as tpo, weights
from tests.golden.make_golden import loss_of

# one parameter's gradient against the oracle's autograd: |g - ref|, |ref|, cos(g, ref), element count
GradErr = collections.namedtuple("GradErr", "err ref cos numel")


def grad_err(g, ref):
    g, ref = g.detach().double().cpu().reshape(-1), ref.detach().double().reshape(-1)
    gn, rn = float(g.norm()), float(ref.norm())
    cos = float(g @ ref) / (gn * rn) if gn > 0 and rn > 0 else (1.0 if gn == rn else 0.0)
    return GradErr(float((g - ref).norm()), rn, cos, g.numel())


# Per-parameter bound of a bf16-arithmetic backward: EVERY parameter whose reference gradient is above the norm floor must point the
# oracle's way (cosine; tensors of >= 8 elements — for a scalar the cosine is just the sign) or be close in norm (relative error), and no
# checked parameter may be further than `worst_max` — a median over the parameters lets one wrong gradient through.
# The floor is relative to the largest per-element gradient RMS of the model: a tensor whose true gradient is three orders of magnitude
# below the others' (a conv bias in front of BatchNorm: mathematically zero; a bias whose contributions cancel) carries the rounding noise
# of the tensors it is summed from.  Calibrated on MI355X (profiles/r06_grad_dist_a.log; the step is bitwise reproducible, so these are
# not noisy): x3f (fp32-class forward, bf16 backward) worst relative error 0.05 on the miniatures and 0.12 at full size (one scalar
# bias of the cross-task MLP), worst cosine 0.9994; bf16 (forward AND backward bf16: the activations differ too) 0.29 / 0.30, cosine 0.968.
PER_PARAM = {"x3f": dict(cos_min=0.999, rel_max=0.15, worst_max=0.2, floor=1e-3),
             "bf16": dict(cos_min=0.95, rel_max=0.35, worst_max=0.5, floor=1e-3)}


def per_param_violations(errs, mode="x3f"):
    """-> (violations [(name, rel, cos, rms / top_rms)], n_checked, n_below_floor) under PER_PARAM[mode]: a parameter above the floor passes
    when (cos >= cos_min and numel >= 8) or rel <= rel_max, and needs rel < worst_max."""
    b = PER_PARAM[mode]
    rms = {k: v.ref / max(v.numel, 1) ** 0.5 for k, v in errs.items()}
    top = max(rms.values())
    bad, checked, below = [], 0, 0
    for k, v in errs.items():
        if rms[k] < b["floor"] * top:
            below += 1
            continue
        checked += 1
        rel = v.err / v.ref
        if not (((v.cos >= b["cos_min"] and v.numel >= 8) or rel <= b["rel_max"]) and rel < b["worst_max"]):
            bad.append((k, rel, v.cos, rms[k] / top))
    return bad, checked, below


def assert_per_param(errs, mode, min_checked_frac=0.6):
    """the per-parameter gradient bound of `mode`; also refuses a floor that would exempt most of the model"""
    bad, checked, below = per_param_violations(errs, mode)
    assert checked >= min_checked_frac * (checked + below), (checked, below)
    assert not bad, (len(bad), bad[:6])
    return checked, below


def grad_errors(name, prec, device, drop=None, seed=0):
    cfg = configs.taskprompter(name)
    model = conftest.build_product_model(cfg, prec, device, drop_path_rate=0.3 if drop is not None else 0.0)
    try:                                   # the state-dict contract dumped from the unmodified reference, where a fixture exists
        contract = conftest.load_golden(cfg.get("contract_of", name))[0]["contract"]
    except OSError:
        contract = [(k, list(v.shape)) for k, v in model.state_dict().items()]
    sd = weights.synth_state_dict(contract, seed)
    model.load_state_dict(sd, strict=True)
    model.train()
    x = weights.synth_images(2, cfg["img_size"], 2)
    if drop is not None:
        model.backbone._drop_override = drop
    out = model(x.to(device))
    loss_of({k: v.cpu() for k, v in out.items()}).backward()
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items() if v.dtype.is_floating_point and "running_" not in k}
    ref_out = tpo.forward(dict(sd, **params), cfg, x, training=True, drop=drop)
    loss_of(ref_out).backward()
    fwd = {t: float((out[t].detach().cpu() - ref_out[t].detach()).norm() / ref_out[t].detach().norm()) for t in ref_out}
    errs = {}
    for k, prm in model.named_parameters():
        ref = params[k].grad if params[k].grad is not None else torch.zeros_like(params[k])
        assert prm.grad is not None, f"{k}: no gradient (reference gives every TaskPrompter parameter one)"
        errs[k] = grad_err(prm.grad, ref)
    return fwd, errs


def skip_unless_own_pitch(cfg):
    """The InvPT TRAINING path needs decoder widths that are their own channel pitch (invpt_autograd._check8).  True for every published
    config and for `mini8` under the default rule; under MTT_TEST_PITCH32_FROM (the miniatures' second run on the wide pitch) it is not."""
    import pytest
    import mtt_amd
    E = cfg["embed_dim"] + cfg["pred_const"]
    dims = (cfg["embed_dim"], E, E // 2, E // 4)
    if any(d != mtt_amd.ops.pad8(d) for d in dims):
        pytest.skip(f"InvPT training path: widths {dims} are not their own channel pitch under PITCH32_FROM = {mtt_amd.ops.PITCH32_FROM}")


def invpt_grad_errors(name, prec, device, seed=0):
    """InvPT: product training forward + backward vs the oracle's autograd (dead reference parameters must get no gradient)."""
    from oracle import invpt_oracle as ipo
    cfg = configs.invpt(name)
    skip_unless_own_pitch(cfg)
    model = conftest.build_product_model(cfg, prec, device)
    contract = [(k, list(v.shape)) for k, v in model.state_dict().items()]
    sd = weights.synth_state_dict(contract, seed)
    model.load_state_dict({k: v.to(device) for k, v in sd.items()}, strict=True)
    model.train()
    x = weights.synth_images(2, cfg["img_size"], 2)
    out = model(x.to(device))
    cpu = {k: v.cpu() for k, v in out.items() if k != "inter_preds"}
    cpu["inter_preds"] = {k: v.cpu() for k, v in out["inter_preds"].items()}
    loss_of(cpu).backward()
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items() if v.dtype.is_floating_point and "running_" not in k}
    ref_out = ipo.forward(dict(sd, **params), cfg, x, training=True)
    loss_of(ref_out).backward()
    fwd = {t: float((cpu[t].detach() - ref_out[t].detach()).norm() / ref_out[t].detach().norm()) for t in cpu if t != "inter_preds"}
    fwd.update({"inter/" + t: float((cpu["inter_preds"][t].detach() - v.detach()).norm() / v.detach().norm())
                for t, v in ref_out["inter_preds"].items()})
    errs, dead = {}, []
    for k, prm in model.named_parameters():
        if params[k].grad is None:
            dead.append(k)
            assert prm.grad is None or float(prm.grad.abs().max()) == 0.0, f"{k}: dead in the reference but got a gradient"
            continue
        assert prm.grad is not None, f"{k}: no gradient"
        errs[k] = grad_err(prm.grad, params[k].grad)
    return fwd, errs, dead


def summarize(errs, floor=1e-6):
    rel = sorted(((v[0] / v[1], k) for k, v in errs.items() if v[1] > floor), reverse=True)
    med = rel[len(rel) // 2][0]
    return rel[0], med


def multi_step_check(name, prec, device, steps=3, lr=5e-3, seed=0):
    """`steps` optimizer steps of the product (FusedClipAdam writes the parameters through raw pointers -> every cached weight pack
    must be rebuilt) vs the oracle driven by clip_grad_norm_ + torch.optim.Adam.  Returns per-step (loss_product, loss_oracle) and
    the final worst relative parameter difference."""
    import mtt_amd
    cfg = configs.taskprompter(name)
    meta, _ = conftest.load_golden(name)
    sd = weights.synth_state_dict(meta["contract"], seed)
    model = conftest.build_product_model(cfg, prec, device)
    model.load_state_dict(sd, strict=True)
    model.train()
    opt = mtt_amd.optim.FusedClipAdam(model.parameters(), lr=lr, weight_decay=1e-6, max_norm=1.0)
    ref = {k: v.clone() for k, v in sd.items()}
    params = {k: ref[k].requires_grad_(True) for k in ref if ref[k].dtype.is_floating_point and "running_" not in k}
    ropt = torch.optim.Adam(list(params.values()), lr=lr, weight_decay=1e-6)
    x = weights.synth_images(2, cfg["img_size"], 2)
    losses = []
    for it in range(steps):
        out = model(x.to(device))
        lp = loss_of({k: v.cpu() for k, v in out.items()})
        opt.zero_grad(set_to_none=True)
        lp.backward()
        opt.step()
        rout = tpo.forward(ref, cfg, x, training=True)
        lr_ = loss_of(rout)
        ropt.zero_grad(set_to_none=True)
        lr_.backward()
        for q in params.values():                      # the reference gives every TaskPrompter parameter a gradient
            if q.grad is None:
                q.grad = torch.zeros_like(q)
        torch.nn.utils.clip_grad_norm_(list(params.values()), 1.0)
        ropt.step()
        losses.append((float(lp.detach()), float(lr_.detach())))
    worst = 0.0
    for k, prm in model.named_parameters():
        d = float((prm.detach().cpu() - params[k].detach()).norm() / params[k].detach().norm().clamp_min(1e-12))
        worst = max(worst, d)
    return losses, worst


def swin_drop_masks(cfg, B, rate=0.3, seed=5):
    """One [4, B] mask / keep table per block (the 4 DropPath draws), shared by the product (model._drop_override) and the oracle."""
    g = torch.Generator().manual_seed(seed)
    keep = 1.0 - rate
    return {(il, ib): torch.bernoulli(torch.full((4, B), keep), generator=g) / keep for il, d in enumerate(cfg["depths"]) for ib in range(d)}


def swin_grad_errors(name, prec, device, seed=0, drop=None, batch=2, contract=None, ref_cache=None):
    """TaskPrompter-Swin: product training forward + backward (swin_autograd.py) vs the oracle's autograd (itself pinned against the
    reference's gradient norms in tests/test_oracle_golden.py).  Parameters the reference leaves without a gradient must stay so.
    contract: the state-dict contract [(name, shape)]; default = the one dumped from the unmodified reference for the miniatures (tests/golden)."""
    from oracle import swin_oracle as swo
    cfg = configs.swin(name)
    model = conftest.build_product_model(cfg, prec, device, drop_path_rate=0.3 if drop is not None else 0.0)
    if contract is None:
        try:
            contract = conftest.load_golden(name)[0]["contract"]
        except OSError:                    # a miniature without a reference fixture: the product's own contract
            contract = [(k, list(v.shape)) for k, v in model.state_dict().items() if k.rsplit(".", 1)[-1] not in weights.DERIVED_BUFFERS]
    sd = weights.synth_state_dict(contract, seed)
    model.load_state_dict({k: v.to(device) for k, v in sd.items()}, strict=False)
    model.train()
    x = weights.synth_images(batch, cfg["img_size"], 2)
    if drop is not None:
        model.backbone._drop_override = drop
    out = model(x.to(device))
    loss_of({k: v.cpu() for k, v in out.items()}).backward()
    if ref_cache is not None and "ref" in ref_cache:                # the oracle's step is the same for every arithmetic mode of the product
        params, ref_out = ref_cache["ref"]
    else:
        params = {k: v.clone().requires_grad_(True) for k, v in sd.items() if v.dtype.is_floating_point and "running_" not in k}
        ref_out = swo.forward(dict(sd, **params), cfg, x, training=True, drop=drop)
        loss_of(ref_out).backward()
        ref_out = {k: v.detach() for k, v in ref_out.items()}
        if ref_cache is not None:
            ref_cache["ref"] = (params, ref_out)
    fwd = {t: float((out[t].detach().cpu() - ref_out[t].detach()).norm() / ref_out[t].detach().norm()) for t in ref_out}
    errs, dead = {}, []
    for k, prm in model.named_parameters():
        rg = params[k].grad
        if rg is None or float(rg.abs().max()) == 0.0:
            dead.append(k)
            assert prm.grad is None or float(prm.grad.abs().max()) < 1e-12, f"{k}: dead in the reference but got a gradient"
            continue
        assert prm.grad is not None, f"{k}: no gradient"
        errs[k] = grad_err(prm.grad, rg)
    return fwd, errs, dead


def trajectory_check(name, modes, device, steps=200, lr=2e-4, n_batches=4, batch=2, seed=0):
    """`steps` optimizer steps of the whole training iteration — forward (train-mode BatchNorm), the reference's criterion
    (MultiTaskLoss; the product uses its fused HIP form), backward, clip_grad_norm_(10) + Adam (train_utils.py:32-51) — from the same
    state on the same cycle of `n_batches` synthetic batches: the CPU oracle (fp32 autograd, torch.optim.Adam) and the product in every
    arithmetic mode of `modes`.  Returns {"oracle": [loss per step], mode: [...]}: what bounds how far mixed-precision training
    (x3f: fp32-class forward, bf16 backward) drifts from the reference's fp32 training over a few hundred steps."""
    import mtt_amd
    cfg = configs.taskprompter(name)
    meta, _ = conftest.load_golden(name)
    sd = weights.synth_state_dict(meta["contract"], seed)
    H, W = cfg["img_size"]
    xs = [weights.synth_images(batch, cfg["img_size"], 20 + i) for i in range(n_batches)]
    curves = {}
    p = None
    for mode in modes:
        model = conftest.build_product_model(cfg, mode, device)
        model.load_state_dict(sd, strict=True)
        model.train()
        p = model.backbone.p
        crit = mtt_amd.losses.FusedMultiTaskLoss(p, p.TASKS.NAMES).to(device)
        gts = [mtt_amd.losses.synthetic_targets(p, batch, H, W, device, seed=30 + i) for i in range(n_batches)]
        opt = mtt_amd.optim.FusedClipAdam(model.parameters(), lr=lr, weight_decay=1e-6, max_norm=10.0)
        xd = [x.to(device) for x in xs]
        ls = []
        for it in range(steps):
            loss = crit(model(xd[it % n_batches]), gts[it % n_batches])["total"]
            opt.zero_grad(set_to_none=True)
            loss.backward()
            opt.step()
            ls.append(loss.detach())
        curves[mode] = [float(v) for v in torch.stack(ls).cpu()]
        del model, opt, crit
        mtt_amd.ops.clear_pack_cache()
    ref = {k: v.clone() for k, v in sd.items()}
    params = {k: ref[k].requires_grad_(True) for k in ref if ref[k].dtype.is_floating_point and "running_" not in k}
    ropt = torch.optim.Adam(list(params.values()), lr=lr, weight_decay=1e-6)
    rcrit = losses_oracle.MultiTaskLoss(p, p.TASKS.NAMES)
    rgts = [mtt_amd.losses.synthetic_targets(p, batch, H, W, "cpu", seed=30 + i) for i in range(n_batches)]
    ls = []
    for it in range(steps):
        # BatchNorm running statistics are state too: the oracle returns the updated buffers through `ref` (in place)
        loss = rcrit(tpo.forward(ref, cfg, xs[it % n_batches], training=True), rgts[it % n_batches])["total"]
        ropt.zero_grad(set_to_none=True)
        loss.backward()
        for q in params.values():
            if q.grad is None:
                q.grad = torch.zeros_like(q)
        torch.nn.utils.clip_grad_norm_(list(params.values()), 10.0)
        ropt.step()
        ls.append(float(loss.detach()))
    curves["oracle"] = ls
    return curves


def trajectory_gaps(curves, window=10):
    """per mode: (max over steps of |loss - oracle| / oracle, the same on `window`-step running means, final-window relative gap)"""
    import numpy as np
    o = np.asarray(curves["oracle"])
    res = {}
    for m, c in curves.items():
        if m == "oracle":
            continue
        c = np.asarray(c)
        k = np.ones(window) / window
        os_, cs_ = np.convolve(o, k, "valid"), np.convolve(c, k, "valid")
        res[m] = (float(np.max(np.abs(c - o) / np.abs(o))), float(np.max(np.abs(cs_ - os_) / np.abs(os_))), float(abs(cs_[-1] - os_[-1]) / abs(os_[-1])))
    return res
