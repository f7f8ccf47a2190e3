"""Generate the criterion / LR-schedule fixtures from the UNMODIFIED reference (run in the build container only).

    python tests/golden/make_loss_golden.py

Imports `losses/loss_functions.py` and `losses/loss_schemes.py` of both sub-projects straight from /root/reference (they
need only torch + numpy), builds the per-task criteria exactly as `utils/common_config.get_loss` does
(TaskPrompter/utils/common_config.py:200-228) and the reference `MultiTaskLoss` (TaskPrompter/losses/loss_schemes.py:9-39,
InvPT/losses/loss_schemes.py:9-33 incl. intermediate supervision), and writes, for seeded logits / labels that are stored
in the fixture too: every task loss, the weighted total and d total / d logits.  `PolynomialLR` is executed from the
class's own source lines in utils/train_utils.py (the module itself imports imageio / evaluation code that is not
available offline, so only the class statement is compiled).

    losses.npz : inputs (pred/<task>, inter/<task>, gt/<case>/<task>) + outputs (<scheme>/<case>/loss/<key>,
                 <scheme>/<case>/grad/<task>, <scheme>/<case>/gradinter/<task>) + lr/<variant>
"""
import ast
import importlib
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import ref_import  # noqa: E402

TASKS = ["semseg", "depth", "human_parts", "sal", "normals", "edge"]
NOUT = dict(semseg=21, human_parts=7, sal=2, normals=3, edge=1, depth=1)
WEIGHTS = dict(semseg=1.0, human_parts=2.0, sal=5.0, edge=50.0, normals=10.0, depth=1.0)     # pascal yml:44-50 (+depth 1.0, nyud yml)
B, H, W = 2, 12, 10


def inputs():
    """Seeded logits and three label cases: 'plain' (5 % ignore regions), 'dense' (no ignored pixel), 'void' (image 1 fully ignored)."""
    g = torch.Generator().manual_seed(11)
    pred = {t: torch.randn(B, NOUT[t], H, W, generator=g) * 2 for t in TASKS}
    inter = {t: torch.randn(B, NOUT[t], H, W, generator=g) * 2 for t in TASKS}
    cases = {}
    for case in ("plain", "dense", "void"):
        gt = {}
        for t in TASKS:
            ign = torch.rand(B, 1, H, W, generator=g) < (0.0 if case == "dense" else 0.05)
            if case == "void":
                ign[1] = True
            if t in ("semseg", "human_parts"):
                y = torch.randint(0, NOUT[t], (B, 1, H, W), generator=g).float()
                y[ign] = 255
            elif t == "sal":
                y = torch.randint(0, 2, (B, 1, H, W), generator=g).float()
                y[ign] = 255
            elif t == "edge":
                y = (torch.rand(B, 1, H, W, generator=g) < 0.1).float()
                y[ign] = 255
            elif t == "normals":
                y = torch.nn.functional.normalize(torch.randn(B, 3, H, W, generator=g), dim=1)
                y[ign.expand(B, 3, H, W)] = 255
            else:
                y = torch.rand(B, 1, H, W, generator=g) * 9.9 + 0.1
                y[ign] = -1
            gt[t] = y
        cases[case] = gt
    return pred, inter, cases


def _import_losses(which):
    root = os.path.join(ref_import.REFERENCE_ROOT, {"TP": "TaskPrompter", "IP": "InvPT"}[which])
    ref_import._purge()
    sys.path.insert(0, root)
    try:
        lf = importlib.import_module("losses.loss_functions")
        ls = importlib.import_module("losses.loss_schemes")
        return lf, ls
    finally:
        sys.path.remove(root)
        ref_import._purge()


def reference_criterion(which, intermediate):
    lf, ls = _import_losses(which)
    ED = ref_import.easydict()
    p = ED(ignore_index=255, edge_w=0.95, intermediate_supervision=intermediate, ignore_invalid_area_depth=True)
    p.TASKS = ED(NAMES=list(TASKS))
    ft = {}
    for t in TASKS:                                      # utils/common_config.py:200-228, verbatim argument choices
        if t == "edge":
            ft[t] = lf.BalancedBinaryCrossEntropyLoss(pos_weight=p["edge_w"], ignore_index=p.ignore_index)
        elif t in ("semseg", "human_parts"):
            ft[t] = lf.CrossEntropyLoss(ignore_index=p.ignore_index)
        elif t == "normals":
            ft[t] = lf.L1Loss(normalize=True, ignore_index=p.ignore_index)
        elif t == "sal":
            ft[t] = lf.CrossEntropyLoss(balanced=True, ignore_index=p.ignore_index)
        elif t == "depth":
            ft[t] = lf.L1Loss(ignore_invalid_area=True, ignore_index=-1) if which == "TP" else lf.L1Loss(ignore_index=-1)
    return ls.MultiTaskLoss(p, list(TASKS), torch.nn.ModuleDict(ft), dict(WEIGHTS))


def reference_polynomial_lr():
    src = open(os.path.join(ref_import.REFERENCE_ROOT, "TaskPrompter", "utils", "train_utils.py")).read()
    node = next(n for n in ast.parse(src).body if isinstance(n, ast.ClassDef) and n.name == "PolynomialLR")
    ns = {"torch": torch}
    exec(compile(ast.Module(body=[node], type_ignores=[]), "train_utils.py:PolynomialLR", "exec"), ns)
    return ns["PolynomialLR"]


def main():
    torch.set_num_threads(1)
    pred, inter, cases = inputs()
    arrays = {}
    for t in TASKS:
        arrays[f"pred/{t}"] = pred[t].numpy()
        arrays[f"inter/{t}"] = inter[t].numpy()
        for case, gt in cases.items():
            arrays[f"gt/{case}/{t}"] = gt[t].numpy()
    for scheme, which, intermediate in (("tp", "TP", False), ("ip", "IP", True)):
        crit = reference_criterion(which, intermediate)
        for case, gt in cases.items():
            pr = {t: pred[t].clone().requires_grad_(True) for t in TASKS}
            arg = dict(pr)
            if intermediate:
                it = {t: inter[t].clone().requires_grad_(True) for t in TASKS}
                arg["inter_preds"] = it
            out = crit(arg, gt, list(TASKS))
            out["total"].backward()
            for k, v in out.items():
                arrays[f"{scheme}/{case}/loss/{k}"] = np.float64(float(v))
            for t in TASKS:
                arrays[f"{scheme}/{case}/grad/{t}"] = pr[t].grad.numpy()
                if intermediate:
                    arrays[f"{scheme}/{case}/gradinter/{t}"] = it[t].grad.numpy()
    PolyLR = reference_polynomial_lr()
    for tag, kw in (("default", dict(max_iterations=40, gamma=0.9, min_lr=0.0)), ("minlr", dict(max_iterations=25, gamma=0.7, min_lr=1e-6))):
        w = torch.nn.Parameter(torch.zeros(1))
        opt = torch.optim.SGD([w], lr=2e-5)
        sch = PolyLR(opt, **kw)
        lrs = []
        for _ in range(kw["max_iterations"] + 1):         # the reference steps the scheduler once per iteration (train_utils.py:51)
            lrs.append(opt.param_groups[0]["lr"])
            opt.step()
            if len(lrs) <= kw["max_iterations"]:
                sch.step()
        arrays[f"lr/{tag}"] = np.array(lrs, dtype=np.float64)
    np.savez_compressed(os.path.join(HERE, "losses.npz"), **arrays)
    print("wrote losses.npz:", len(arrays), "arrays,", {k: float(v) for k, v in arrays.items() if "/loss/" in k and "plain" in k})


if __name__ == "__main__":
    main()
