"""Micro-benchmark of mtt_gemm variants on the real shapes of the NS-6 training step (TFLOP/s per shape and variant)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import mtt_amd  # noqa: E402
from mtt_amd import ops  # noqa: E402

_call = ops.call
FORCE = {"v": 0}
ops.call = lambda name, **kw: _call(name, **(dict(kw, variant=FORCE["v"]) if name == "gemm" else kw))     # mtt_gemm_desc.variant
prec = ops.Prec("bf16")
SHAPES = [("qkv", 8240, 3072, 1024), ("proj", 8240, 1024, 1024), ("fc1", 8240, 4096, 1024), ("fc2", 8240, 1024, 4096),
          ("big", 8192, 8192, 8192), ("fc1_b16", 16480, 4096, 1024)]


def bench(M, N, K, variant, iters=20):
    FORCE["v"] = variant
    x = torch.randn(M, K, device="cuda").bfloat16()
    w = torch.randn(1, N, K, device="cuda").bfloat16()
    out = torch.empty(1, M, N, device="cuda", dtype=torch.bfloat16)
    for _ in range(3):
        ops.linear(x, w, N, prec, out=out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        ops.linear(x, w, N, prec, out=out)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    return 2.0 * M * N * K / ms / 1e9, ms


for name, M, N, K in SHAPES:
    row = [f"{name:8s} M={M:6d} N={N:5d} K={K:5d}"]
    for v, vn in ((1, "reg128"), (2, "dma128"), (3, "dma256")):
        tf, ms = bench(M, N, K, v)
        row.append(f"{vn} {tf:7.1f} TF/s ({ms * 1e3:7.1f} us)")
    print("  ".join(row), flush=True)
