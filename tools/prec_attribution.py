"""Per-group error attribution of the bf16 mode: full-size forward of a BASELINE config through the HIP kernels
against the CPU oracle, with bf16 operand rounding in exactly one group of the path (and the converse: x3 in exactly one group).

    python tools/prec_attribution.py [--config ns6] [--batch 2] [--out gpurun_out/r03_prec_attribution.json]

Groups (taskprompter.PREC_GROUPS): enc = patch embed + qkv / proj / fc1 / fc2 Linears, attn = QK^T / softmax / PV, side = token_trans(1) +
channel logits + fea_decode_{spa,chan} on the modulated features, fuse = fea_fuse convs, heads = ConvHead / DEConvHead.
Storage is fp32 in every run, so a row isolates OPERAND rounding (which is all the bf16 mode adds: its residual stream, statistics and
logit side channels are fp32 already).  `all bf16` must reproduce the bf16 mode's own error (checked in the output)."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="ns6")
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "r03_prec_attribution.json"))
    a = ap.parse_args()
    import conftest
    import parity_util as pu
    import mtt_amd
    from oracle import configs
    torch.set_num_threads(conftest.HOST_THREADS)
    cfg, sd, x, ref = pu.oracle_eval(a.config, a.batch)
    G = mtt_amd.taskprompter.PREC_GROUPS
    C, depth, nH, sel = configs.VIT[cfg["backbone"]]

    def run(prec, groups=None):
        mtt_amd.ops.clear_pack_cache()
        extra = dict(mtt_prec_groups=groups) if groups else {}
        p = mtt_amd.factory.make_p([t for t, _ in cfg["tasks"]], cfg["img_size"], backbone=(C, depth, nH, sel), head=cfg["head"],
                                   embed_dim=cfg["embed_dim"], final_embed_dim=cfg["final_embed_dim"], chan_nheads=cfg["chan_nheads"],
                                   use_ctr=cfg["use_ctr"], num_output=dict(cfg["tasks"]), prec=prec, drop_path_rate=0.0, **extra)
        m = mtt_amd.factory.get_model(p).cuda()
        m.load_state_dict(sd)
        m.eval()
        with torch.no_grad():
            out = m(x.cuda())
            torch.cuda.synchronize()
        e = pu.head_errors(out, ref)
        del m
        torch.cuda.empty_cache()
        return e

    rows = [("x3 everywhere", run("x3")), ("bf16 mode (bf16 storage)", run("bf16")),
            ("all groups bf16 operands (fp32 storage)", run("x3", {g: "bf16" for g in G}))]
    for g in G:
        rows.append((f"bf16 only in {g}", run("x3", {g: "bf16"})))
    for g in G:
        rows.append((f"x3 only in {g}", run("x3", {k: ("x3" if k == g else "bf16") for k in G})))
    rec = dict(config=a.config, batch=a.batch, tolerance=1e-3, metric="per-head relative L2 error vs the CPU oracle",
               rows=[dict(run=n, worst=max(e.values()), per_head=e) for n, e in rows])
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    json.dump(rec, open(a.out, "w"), indent=1)
    w = max(len(n) for n, _ in rows)
    print(f"{'run':{w}}  worst      " + "  ".join(f"{t:>11}" for t in rows[0][1]))
    for n, e in rows:
        print(f"{n:{w}}  {max(e.values()):.3e}  " + "  ".join(f"{v:11.3e}" for v in e.values()))


if __name__ == "__main__":
    main()
