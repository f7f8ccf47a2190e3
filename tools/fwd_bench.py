"""Forward-only timing of a TaskPrompter config on the GPU (inference path), for profiling.
Usage: python tools/fwd_bench.py [--batch 8] [--iters 5] [--backbone TaskPrompter_vitL] [--prec bf16]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import mtt_amd  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--backbone", default="TaskPrompter_vitL")
    ap.add_argument("--prec", default="bf16")
    a = ap.parse_args()
    p = mtt_amd.factory.make_p(mtt_amd.factory.TASK_ORDER, (512, 512), backbone=a.backbone, prec=a.prec)
    model = mtt_amd.factory.get_model(p).cuda().eval()
    x = torch.randn(a.batch, 3, 512, 512, device="cuda")
    with torch.no_grad():
        for _ in range(a.warmup):
            model(x)
        torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(a.iters):
            model(x)
        torch.cuda.synchronize()
    ms = (time.time() - t0) / a.iters * 1e3
    print(f"fwd {a.backbone} {a.prec} B={a.batch}: {ms:.2f} ms/batch, {ms / a.batch:.3f} ms/img, "
          f"{1046.8 * a.batch / ms:.1f} TFLOP/s (NS-6 algorithmic 1046.8 GF/img)")


if __name__ == "__main__":
    main()
