"""Full-model forward parity on the GPU against the golden fixtures (and quick timing of the NS config).
Writes gpurun_out/model_check.json.  Usage: python tools/gpu_model_check.py [--time]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

import conftest  # noqa: E402
from oracle import configs, weights  # noqa: E402


def main():
    rep = {}
    dev = "cuda"
    for name in ("mini_ctr", "mini_win", "mini_deconv"):
        cfg = configs.taskprompter(name)
        meta, gold = conftest.load_golden(name)
        for prec in ("x3", "bf16"):
            try:
                model = conftest.build_product_model(cfg, prec, dev)
                model.load_state_dict(weights.synth_state_dict(meta["contract"], 0), strict=True)
                model.eval()
                x = weights.synth_images(meta["batch"], cfg["img_size"], 1).to(dev)
                with torch.no_grad():
                    out = model(x)
                torch.cuda.synchronize()
                errs = {}
                for t in out:
                    g = torch.from_numpy(gold[f"eval/{t}"])
                    errs[t] = float((out[t].cpu() - g).norm() / g.norm())
                rep[f"{name}/{prec}/eval"] = errs
                print(name, prec, "eval", {k: f"{v:.2e}" for k, v in errs.items()}, flush=True)
                model.train()
                x2 = weights.synth_images(2, cfg["img_size"], 2).to(dev)
                with torch.no_grad():
                    out = model(x2)
                torch.cuda.synchronize()
                errs = {}
                for t in out:
                    g = torch.from_numpy(gold[f"train/{t}"])
                    errs[t] = float((out[t].cpu()[:, :, ::2, ::2] - g).norm() / g.norm())
                rep[f"{name}/{prec}/train_fwd"] = errs
                print(name, prec, "train_fwd", {k: f"{v:.2e}" for k, v in errs.items()}, flush=True)
            except Exception as e:  # noqa: BLE001
                import traceback
                traceback.print_exc()
                rep[f"{name}/{prec}"] = repr(e)
    if "--time" in sys.argv:
        import mtt_amd
        for bname, B in (("TaskPrompter_vitL", 4), ("TaskPrompter_vitL", 8)):
            p = mtt_amd.factory.make_p(mtt_amd.factory.TASK_ORDER, (512, 512), backbone=bname, prec="bf16")
            model = mtt_amd.factory.get_model(p).to(dev).eval()
            x = torch.randn(B, 3, 512, 512, device=dev)
            with torch.no_grad():
                for _ in range(2):
                    model(x)
                torch.cuda.synchronize()
                t0 = time.time()
                n = 5
                for _ in range(n):
                    model(x)
                torch.cuda.synchronize()
            ms = (time.time() - t0) / n * 1e3
            rep[f"time/ns6/B{B}"] = dict(ms_per_batch=ms, ms_per_img=ms / B)
            print("NS6 fwd bf16 B", B, f"{ms:.2f} ms/batch {ms / B:.2f} ms/img", flush=True)
            del model
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(rep, open(os.path.join(ROOT, "gpurun_out", "model_check.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
