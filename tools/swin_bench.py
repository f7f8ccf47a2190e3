"""Forward (inference) throughput of the TaskPrompter-Swin path: Swin-B, window 12, 1024x2048 x 0.75 (cs_swinB_taskprompter.yml without
the 3ddet task), DEConv heads, bf16 mode; images / s and ms per image for a few batch sizes, plus the kernel-time share of the new
kernels when run under rocprofv3 (tools/prof_summary.py).  Weights: the deterministic synthetic state dict."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

import conftest  # noqa: E402
from oracle import configs, weights  # noqa: E402  (config table + synthetic weights only; no oracle arithmetic is timed)

cfg = configs.swin("cs_swinB")
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
batches = [int(b) for b in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1, 4]
train = len(sys.argv) > 3 and sys.argv[3] == "train"      # forward + surrogate loss + backward + clip + Adam (drop_path 0)
model = conftest.build_product_model(cfg, prec, "cuda", drop_path_rate=0.0)
contract = [(k, list(v.shape)) for k, v in model.state_dict().items()]
model.load_state_dict({k: v.cuda() for k, v in weights.synth_state_dict(contract, 0).items()}, strict=False)
model.eval()
for B in batches:
    x = weights.synth_images(B, cfg["img_size"], 1).cuda()
    with torch.no_grad():
        for _ in range(2):
            out = model(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 5
        for _ in range(n):
            out = model(x)
        torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f"swinB {prec} B={B}: {dt * 1e3 / B:8.2f} ms / image  {B / dt:7.2f} images/s  peak HBM {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB", flush=True)

if train:
    import mtt_amd
    model.train()
    opt = mtt_amd.optim.FusedClipAdam(model.parameters(), lr=2e-5, max_norm=10.0)
    for B in batches:
        x = weights.synth_images(B, cfg["img_size"], 1).cuda()
        g = torch.Generator(device="cuda").manual_seed(3)
        rnd = {t: torch.randn(B, n, *cfg["img_size"], device="cuda", generator=g) for t, n in cfg["tasks"]}

        def step():
            out = model(x)
            loss = sum((out[t] * rnd[t]).mean() for t in out)
            opt.zero_grad(set_to_none=True)
            loss.backward()
            opt.step()
            return loss
        for _ in range(2):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 3
        for _ in range(n):
            loss = step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        print(f"swinB {prec} TRAIN B={B}: {dt * 1e3:8.1f} ms / step  {B / dt:6.2f} images/s  loss {float(loss):.4f}  peak HBM {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB",
              flush=True)
