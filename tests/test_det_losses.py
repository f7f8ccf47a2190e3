"""Detection-branch losses (det_losses.py FocalLoss / SmoothL1Loss) against the fixture produced by the UNMODIFIED reference
file (tests/golden/make_detloss_golden.py: the reference's own host path, py_sigmoid_focal_loss):
  * CPU: the oracle restatement (oracle/detloss_oracle.py) and the product modules on the ABI emulator,
  * -m gpu: the product modules on the HIP kernels mtt_detloss_fwd / mtt_detloss_bwd (values, gradients, bitwise repeatability)."""
import os

import numpy as np
import pytest
import torch

import conftest
from tests.golden.make_detloss_golden import CASES


def _gold():
    return np.load(os.path.join(conftest.GOLDEN, "detloss.npz"))


def _case_tensors(g, name, device):
    pred = torch.from_numpy(g[f"{name}/pred"]).to(device).requires_grad_(True)
    target = torch.from_numpy(g[f"{name}/target"]).to(device)
    weight = torch.from_numpy(g[f"{name}/weight"]).to(device) if f"{name}/weight" in g.files else None
    return pred, target, weight, torch.from_numpy(g[f"{name}/up"]).to(device)


def _check(loss, pred, up, g, name, tol, gtol):
    (loss * up).sum().backward()
    ref_l, ref_g = torch.from_numpy(g[f"{name}/loss"]), torch.from_numpy(g[f"{name}/grad"])
    assert loss.shape == ref_l.shape, (name, loss.shape, ref_l.shape)
    assert float((loss.detach().cpu() - ref_l).abs().max()) <= tol * max(1.0, float(ref_l.abs().max())), (name, loss, ref_l)
    assert float((pred.grad.cpu() - ref_g).norm()) <= gtol * float(ref_g.norm()) + 1e-12, name


def test_oracle_restatement_matches_reference():
    from oracle import detloss_oracle as dlo
    g = _gold()
    for name, kind, N, C, wmode, reduction, avg, lw, extra in CASES:
        pred, target, weight, up = _case_tensors(g, name, "cpu")
        if kind == "focal":
            loss = dlo.focal_loss(pred, target, weight, extra["gamma"], extra["alpha"], reduction, avg, lw)
        else:
            loss = dlo.smooth_l1(pred, target, weight, extra["beta"], reduction, avg, lw)
        _check(loss, pred, up, g, name, 1e-6, 2e-6)


def _product(device, tol, gtol):
    import mtt_amd
    g = _gold()
    for name, kind, N, C, wmode, reduction, avg, lw, extra in CASES:
        pred, target, weight, up = _case_tensors(g, name, device)
        if kind == "focal":
            crit = mtt_amd.det_losses.FocalLoss(use_sigmoid=True, gamma=extra["gamma"], alpha=extra["alpha"], reduction=reduction, loss_weight=lw)
        else:
            crit = mtt_amd.det_losses.SmoothL1Loss(beta=extra["beta"], reduction=reduction, loss_weight=lw)
        loss = crit(pred, target, weight, avg_factor=avg)
        _check(loss, pred, up, g, name, tol, gtol)


def test_product_modules_on_the_emulator_match_reference(emulated):
    _product("cpu", 2e-6, 1e-5)


def test_reduction_rules_of_the_reference():
    """weight_reduce_loss's error rule (det_losses.py:50-52) and the empty-target shortcut of smooth_l1_loss (:117-118)."""
    import mtt_amd
    with pytest.raises(ValueError):
        mtt_amd.det_losses.SmoothL1Loss(reduction="sum")(torch.zeros(2, 2), torch.ones(2, 2), avg_factor=3.0)
    z = mtt_amd.det_losses.SmoothL1Loss()(torch.zeros(0, 4, requires_grad=True), torch.zeros(0, 4))
    assert float(z.detach()) == 0.0


@pytest.mark.gpu
def test_product_modules_on_the_device_match_reference():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    _product("cuda", 1e-5, 5e-5)


@pytest.mark.gpu
def test_focal_loss_at_fcos3d_scale_is_deterministic_and_matches_the_oracle():
    """FCOS3D's classification loss at its real size (det_head.py: 5 FPN levels of a 1024 x 2048 image at strides 8..128 = 349 k points,
    10 classes; avg_factor = number of positives): the HIP sum is bitwise repeatable (fixed-order reduction) and equals the oracle's."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import mtt_amd
    from oracle import detloss_oracle as dlo
    g = torch.Generator().manual_seed(3)
    N, C = 349184, 10
    pred = torch.randn(N, C, generator=g) * 3 - 2
    labels = torch.where(torch.rand(N, generator=g) < 0.003, torch.randint(0, C, (N,), generator=g), torch.full((N,), C))
    npos = float((labels < C).sum())
    ref_p = pred.clone().requires_grad_(True)
    ref = dlo.focal_loss(ref_p, labels, None, 2.0, 0.25, "mean", npos, 1.0)
    ref.backward()
    crit = mtt_amd.det_losses.FocalLoss()
    vals = []
    for _ in range(2):
        p = pred.cuda().requires_grad_(True)
        loss = crit(p, labels.cuda(), avg_factor=npos)
        loss.backward()
        vals.append((loss.detach().clone(), p.grad.clone()))
    assert torch.equal(vals[0][0], vals[1][0]) and torch.equal(vals[0][1], vals[1][1])
    assert abs(float(vals[0][0]) - float(ref)) <= 2e-5 * abs(float(ref))
    assert float((vals[0][1].cpu() - ref_p.grad).norm()) <= 2e-5 * float(ref_p.grad.norm())
