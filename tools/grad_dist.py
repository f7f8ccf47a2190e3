"""Per-parameter gradient error distribution of a bf16-arithmetic backward against the oracle's autograd: what the per-parameter bound of
tests/train_check.py (PER_PARAM) is calibrated on.  python tools/grad_dist.py [--device cuda|cpu] [--out FILE] case:prec ...
case = a TaskPrompter miniature (mini_ctr ...), `mini8` (InvPT) or `ns6` (full size, B = 2); on cpu the C ABI is the emulator."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import torch  # noqa: E402

import conftest  # noqa: E402
import train_check  # noqa: E402


def ns6_errors(prec, device):
    import parity_util as pu
    from oracle import taskprompter_oracle as tpo
    from tests.golden.make_golden import loss_of
    cfg, sd, x, _ = pu.oracle_eval("ns6", 2)
    if "ref" not in _NS6:
        params = {k: v.clone().requires_grad_(True) for k, v in sd.items() if v.dtype.is_floating_point and "running_" not in k}
        ref_out = tpo.forward(dict(sd, **params), cfg, x, training=True)
        loss_of(ref_out).backward()
        _NS6["ref"] = params
    params = _NS6["ref"]
    model = conftest.build_product_model(cfg, prec, device)
    model.load_state_dict(sd, strict=True)
    model.train()
    out = model(x.to(device))
    loss_of({k: v.cpu() for k, v in out.items()}).backward()
    errs = {k: train_check.grad_err(prm.grad, params[k].grad if params[k].grad is not None else torch.zeros_like(params[k]))
            for k, prm in model.named_parameters()}
    del model, out
    return errs


_NS6 = {}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("cases", nargs="+")
    ap.add_argument("--device", default="cuda" if torch.cuda.is_available() else "cpu")
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    torch.set_num_threads(conftest.HOST_THREADS)
    if a.device == "cpu":
        import mtt_amd
        from oracle import abi_emul
        mtt_amd.ops.call = abi_emul.call
    rec = {}
    for c in a.cases:
        name, prec = c.split(":")
        if name == "ns6":
            errs = ns6_errors(prec, a.device)
        elif name == "mini8":
            errs = train_check.invpt_grad_errors(name, prec, a.device)[1]
        else:
            errs = train_check.grad_errors(name, prec, a.device)[1]
        rms = {k: v.ref / v.numel ** 0.5 for k, v in errs.items()}
        top = max(rms.values())
        rows = sorted(((v.err / max(v.ref, 1e-300), v.cos, rms[k] / top, v.numel, k) for k, v in errs.items()), reverse=True)
        rec[c] = [dict(rel=r[0], cos=r[1], rms_frac=r[2], numel=r[3], name=r[4]) for r in rows]
        bad, checked, below = train_check.per_param_violations(errs, "bf16" if prec == "bf16" else "x3f")
        print(f"== {c}: {len(rows)} parameters, {checked} above the floor, {below} below, violations {len(bad)}")
        floor = train_check.PER_PARAM["x3f"]["floor"]
        live = [r for r in rows if r[2] >= floor]
        for r in live[:10]:
            print("   rel %.3e  cos %.6f  rms/top %.2e  n %8d  %s" % r)
        for r in [r for r in rows if 1e-6 <= r[2] < floor][:6]:
            print("   [below floor] rel %.3e  cos %.6f  rms/top %.2e  n %8d  %s" % r)
        for b in bad:
            print("   [VIOLATION] %s rel %.3e cos %.6f rms/top %.2e" % b)
        worst_cos = sorted(live, key=lambda r: r[1])[:6]
        for r in worst_cos:
            print("   [cos] rel %.3e  cos %.6f  rms/top %.2e  n %8d  %s" % r)
        sys.stdout.flush()
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        json.dump(rec, open(a.out, "w"))


if __name__ == "__main__":
    main()
