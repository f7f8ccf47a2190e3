"""Generate the golden fixtures from the UNMODIFIED reference (run in the build container only).

    python tests/golden/make_golden.py

For every miniature config it builds the reference nn.Module (imported from /root/reference via
oracle/ref_import.py), loads deterministic synthetic weights (oracle/weights.py), runs it on a
seeded input on CPU in fp32 and writes
  <name>.json : the state-dict contract [(key, shape)] + meta   (drop-in boundary)
  <name>.npz  : eval outputs, train-mode outputs (batch-stat BN), post-step BN buffers and
                per-parameter gradient statistics (L2 norm, sum) of loss = sum_t mean(out_t * r_t)
The GPU box has no /root/reference: tests read only these files.
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import configs, ref_build, weights  # noqa: E402

CASES = [("TP", "mini_ctr", 1), ("TP", "mini_win", 1), ("TP", "mini_deconv", 2), ("IP", "mini", 1), ("TPS", "mini_swin", 2), ("TPS", "mini_swin_pad", 1)]
EVAL_STRIDE = {"mini": 2, "mini_swin": 2, "mini_swin_pad": 3}      # store every n-th pixel to keep the fixtures small
TRAIN_STRIDE = {"mini_swin": 4, "mini_swin_pad": 6}


def loss_of(out, seed=7):
    g = torch.Generator().manual_seed(seed)
    tot = 0.0
    for k in sorted(k for k in out if k != "inter_preds"):
        tot = tot + (out[k] * torch.randn(out[k].shape, generator=g)).mean()
    if "inter_preds" in out:
        for k in sorted(out["inter_preds"]):
            tot = tot + (out["inter_preds"][k] * torch.randn(out["inter_preds"][k].shape, generator=g)).mean()
    return tot


def main(only=None):
    torch.set_num_threads(max(1, os.cpu_count() or 1))
    for kind, name, batch in CASES:
        if only and name not in only:
            continue
        cfg = configs.taskprompter(name) if kind == "TP" else (configs.swin(name) if kind == "TPS" else configs.invpt(name))
        model, p = ref_build.build_reference(cfg, randomize=False)
        contract = [(k, list(v.shape)) for k, v in model.state_dict().items()]
        sd = weights.synth_state_dict(contract, seed=0, keep=model.state_dict())     # geometry-derived buffers keep the model's own values
        model.load_state_dict(sd, strict=True)
        x = weights.synth_images(batch, cfg["img_size"], seed=1)
        arrays = {}
        model.eval()
        with torch.no_grad():
            out = model(x)
        es = EVAL_STRIDE.get(name, 1)
        for k, v in out.items():
            if k == "inter_preds":
                for kk, vv in v.items():
                    arrays[f"eval/inter/{kk}"] = vv[:, :, ::es, ::es].numpy()
            else:
                arrays[f"eval/{k}"] = v[:, :, ::es, ::es].numpy()
        # train mode: batch statistics in BN, DropPath rate 0 (stochastic depth is not reproducible)
        model.train()
        x2 = weights.synth_images(2, cfg["img_size"], seed=2)
        out = model(x2)
        ts = TRAIN_STRIDE.get(name, 2)
        for k, v in out.items():
            if k == "inter_preds":
                for kk, vv in v.items():
                    arrays[f"train/inter/{kk}"] = vv.detach()[:, :, ::ts, ::ts].numpy()
            else:
                arrays[f"train/{k}"] = v.detach()[:, :, ::ts, ::ts].numpy()   # subsampled: fixture size
        loss_of(out).backward()
        gstat = {}
        for k, prm in model.named_parameters():
            if prm.grad is None:
                gstat[k] = None
            else:
                gstat[k] = [float(prm.grad.double().norm()), float(prm.grad.double().sum())]
        for k, v in model.state_dict().items():
            if k.endswith("running_mean") or k.endswith("running_var"):
                arrays[f"bn/{k}"] = v.numpy()
        meta = dict(kind=kind, name=name, batch=batch, eval_stride=es, train_stride=ts, contract=contract, grad_stats=gstat,
                    torch=torch.__version__)
        with open(os.path.join(HERE, f"{name}.json"), "w") as f:
            json.dump(meta, f)
        np.savez_compressed(os.path.join(HERE, f"{name}.npz"), **{k: v.astype(np.float32) for k, v in arrays.items()})
        print(name, "params", sum(int(np.prod(s)) for _, s in contract),
              "npz bytes", os.path.getsize(os.path.join(HERE, f"{name}.npz")))


if __name__ == "__main__":
    main(sys.argv[1:])
