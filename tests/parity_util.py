"""Shared helpers of the full-size `-m gpu` parity tests: oracle outputs per BASELINE config (computed once per process on the
host cores) and a JSON-lines report of every measured error (gpurun_out/parity_report.jsonl on the GPU box; copied to profiles/)."""
import json
import os
import time

import torch

import conftest
from oracle import configs, weights

_CACHE = {}
REPORT = os.path.join(conftest.ROOT, "gpurun_out", "parity_report.jsonl")


def report(kind, **fields):
    rec = dict(kind=kind, **fields)
    print("PARITY " + json.dumps(rec), flush=True)
    try:
        os.makedirs(os.path.dirname(REPORT), exist_ok=True)
        with open(REPORT, "a") as f:
            f.write(json.dumps(rec) + "\n")
    except OSError:
        pass


def get_cfg(name):
    if "swin" in name:
        return configs.swin(name)
    return configs.invpt(name) if name.startswith("cfg4") or name in ("cfg1", "mini", "mini8") else configs.taskprompter(name)


def contract_of(cfg):
    model = conftest.build_product_model(cfg, "x3", "cpu")
    return [(k, list(v.shape)) for k, v in model.state_dict().items()]


def oracle_eval(name, B, seed=0):
    """(cfg, state_dict, images, oracle eval outputs) for config `name`; cached (the x3 and bf16 cases share it)."""
    key = (name, B, seed)
    if key not in _CACHE:
        cfg = get_cfg(name)
        sd = weights.synth_state_dict(contract_of(cfg), seed)
        x = weights.synth_images(B, cfg["img_size"], seed + 1)
        torch.set_num_threads(conftest.HOST_THREADS)
        t0 = time.time()
        with torch.no_grad():
            if cfg["model"] == "TransformerNet":
                from oracle import invpt_oracle as ipo
                ref = ipo.forward(sd, cfg, x)
            elif cfg["model"] == "TaskPrompterSwin":
                from oracle import swin_oracle as swo
                ref = swo.forward(sd, cfg, x)
            else:
                from oracle import taskprompter_oracle as tpo
                ref = tpo.forward(sd, cfg, x)
        report("oracle_time", config=name, batch=B, seconds=round(time.time() - t0, 1), host_threads=torch.get_num_threads())
        _CACHE[key] = (cfg, sd, x, ref)
    return _CACHE[key]


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def head_errors(out, ref):
    errs = {t: rel(out[t], ref[t]) for t in ref if t != "inter_preds"}
    if "inter_preds" in ref:
        errs.update({"inter/" + t: rel(out["inter_preds"][t], v) for t, v in ref["inter_preds"].items()})
    return errs
