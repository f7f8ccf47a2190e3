"""CPU: the oracle restatement (oracle/taskprompter_oracle.py) against the golden fixtures produced by the
UNMODIFIED reference (tests/golden/make_golden.py).  Pins forward (eval + train-mode BN), the BN running
stat update and — through autograd of the restatement — per-parameter gradient statistics."""
import numpy as np
import pytest
import torch

import conftest
from oracle import configs, taskprompter_oracle as tpo, weights
from tests.golden.make_golden import loss_of

TP_CASES = ["mini_ctr", "mini_win", "mini_deconv"]


@pytest.mark.parametrize("name", TP_CASES)
def test_oracle_forward_eval(name):
    cfg = configs.taskprompter(name)
    meta, gold = conftest.load_golden(name)
    sd = weights.synth_state_dict(meta["contract"], 0)
    x = weights.synth_images(meta["batch"], cfg["img_size"], 1)
    with torch.no_grad():
        out = tpo.forward(sd, cfg, x)
    for t, _ in cfg["tasks"]:
        g = torch.from_numpy(gold[f"eval/{t}"])
        assert float((out[t] - g).norm() / g.norm()) < 2e-5, t


def test_oracle_dd_label_map_size_matches_reference_wrapper():
    """taskprompter_wrapper.py:17-27: with `dd_label_map_size` in the config the predictions are resized to THAT size instead of the
    input's.  Fixture from the unmodified reference wrapper (tests/golden/make_dd_golden.py): mini_ctr, 64 x 96 input -> 40 x 56 maps."""
    import os
    cfg = configs.taskprompter("mini_ctr_dd")
    meta, _ = conftest.load_golden("mini_ctr")
    gold = np.load(os.path.join(conftest.GOLDEN, "mini_ctr_dd.npz"))
    sd = weights.synth_state_dict(meta["contract"], 0)
    x = weights.synth_images(2, cfg["img_size"], 1)
    with torch.no_grad():
        out = tpo.forward(sd, cfg, x)
    for t, _ in cfg["tasks"]:
        g = torch.from_numpy(gold[f"eval/{t}"])
        assert out[t].shape == g.shape and tuple(g.shape[-2:]) == (40, 56)
        assert float((out[t] - g).norm() / g.norm()) < 2e-5, t


@pytest.mark.parametrize("name", TP_CASES)
def test_oracle_train_forward_backward(name):
    cfg = configs.taskprompter(name)
    meta, gold = conftest.load_golden(name)
    sd = weights.synth_state_dict(meta["contract"], 0)
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items() if v.dtype.is_floating_point and "running_" not in k}
    full = dict(sd, **params)
    x = weights.synth_images(2, cfg["img_size"], 2)
    upd = {}
    out = tpo.forward(full, cfg, x, training=True, bn_updates=upd)
    for t, _ in cfg["tasks"]:
        g = torch.from_numpy(gold[f"train/{t}"])
        assert float((out[t].detach()[:, :, ::2, ::2] - g).norm() / g.norm()) < 2e-5, t
    for k, v in upd.items():
        g = torch.from_numpy(gold[f"bn/{k}"])
        assert float((v - g).abs().max()) < 1e-4, k
    loss_of(out).backward()
    bad = []
    for k, st in meta["grad_stats"].items():
        if st is None:
            continue
        gn = float(params[k].grad.double().norm()) if params[k].grad is not None else 0.0
        if abs(gn - st[0]) > 2e-3 * max(st[0], 1e-6) + 1e-7:
            bad.append((k, gn, st[0]))
    assert not bad, bad[:5]


def test_invpt_oracle_matches_reference_golden():
    from oracle import invpt_oracle as ipo
    cfg = configs.invpt("mini")
    meta, gold = conftest.load_golden("mini")
    sd = weights.synth_state_dict(meta["contract"], 0)
    es = meta["eval_stride"]
    with torch.no_grad():
        out = ipo.forward(sd, cfg, weights.synth_images(meta["batch"], cfg["img_size"], 1))
        upd = {}
        out_tr = ipo.forward(sd, cfg, weights.synth_images(2, cfg["img_size"], 2), training=True, bn_updates=upd)
    for t, _ in cfg["tasks"]:
        g = torch.from_numpy(gold[f"eval/{t}"])
        assert float((out[t][:, :, ::es, ::es] - g).norm() / g.norm()) < 2e-5, t
        g = torch.from_numpy(gold[f"eval/inter/{t}"])
        assert float((out["inter_preds"][t][:, :, ::es, ::es] - g).norm() / g.norm()) < 2e-5, t
        g = torch.from_numpy(gold[f"train/{t}"])
        assert float((out_tr[t][:, :, ::2, ::2] - g).norm() / g.norm()) < 2e-5, t
    for k, v in upd.items():
        assert float((v - torch.from_numpy(gold[f"bn/{k}"])).abs().max()) < 1e-4, k


SWIN_CASES = ["mini_swin", "mini_swin_pad"]


@pytest.mark.parametrize("name", SWIN_CASES)
def test_swin_oracle_matches_reference_golden(name):
    """oracle/swin_oracle.py against the unmodified taskprompter_swin.py: eval outputs, train-mode outputs (batch-stat BN), BN buffer
    updates and per-parameter gradient norms through the restatement's autograd (shifted / padded windows, patch merging of features,
    attention maps and prompts, channel attention with 1 and 4 windows, Conv and DEConv heads)."""
    from oracle import swin_oracle as swo
    cfg = configs.swin(name)
    meta, gold = conftest.load_golden(name)
    sd = weights.synth_state_dict(meta["contract"], 0)
    es, ts = meta["eval_stride"], meta["train_stride"]
    with torch.no_grad():
        out = swo.forward(sd, cfg, weights.synth_images(meta["batch"], cfg["img_size"], 1))
    for t, _ in cfg["tasks"]:
        g = torch.from_numpy(gold[f"eval/{t}"])
        assert float((out[t][:, :, ::es, ::es] - g).norm() / g.norm()) < 2e-5, t
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items() if v.dtype.is_floating_point and "running_" not in k}
    upd = {}
    out = swo.forward(dict(sd, **params), cfg, weights.synth_images(2, cfg["img_size"], 2), training=True, bn_updates=upd)
    for t, _ in cfg["tasks"]:
        g = torch.from_numpy(gold[f"train/{t}"])
        assert float((out[t].detach()[:, :, ::ts, ::ts] - g).norm() / g.norm()) < 2e-5, t
    for k, v in upd.items():
        assert float((v - torch.from_numpy(gold[f"bn/{k}"])).abs().max()) < 1e-4, k
    loss_of(out).backward()
    bad = []
    for k, st in meta["grad_stats"].items():
        if st is None:
            assert params[k].grad is None or float(params[k].grad.abs().max()) == 0.0, k
            continue
        gn = float(params[k].grad.double().norm()) if params[k].grad is not None else 0.0
        if abs(gn - st[0]) > 2e-3 * max(st[0], 1e-6) + 1e-7:
            bad.append((k, gn, st[0]))
    assert not bad, bad[:5]
