"""Ablations of the persistent LDS-DMA GEMM (gemm_pdma_kernel, bf16 output + bias) next to the one-tile-per-workgroup kernel: where the
~10 us per 256 x 256 tile that are not K steps go.  Interleaved rounds in one process; prints the median time per launch and per
round of 256 tiles.  Variants 6 / 15 / 16 / 17 are measurement-only (wrong results by construction)."""
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import mtt_amd  # noqa: E402
from mtt_amd import ops  # noqa: E402

prec = ops.Prec("bf16")
M63 = 63 * 1030
SHAPES = [("qkv", M63, 3072, 1024), ("proj", M63, 1024, 1024), ("fc2", M63, 1024, 4096), ("big", 8192, 8192, 8192)]
KERNELS = [(14, "one tile / wg"), (6, "one tile / wg, no epilogue"), (12, "persistent, deferred stores"), (13, "persistent, immediate stores"),
           (15, "persistent, no epilogue"), (16, "persistent, no stores"), (17, "persistent, no bias loads")]
ROUNDS = int(sys.argv[1]) if len(sys.argv) > 1 else 5

for name, M, N, K in SHAPES:
    x = (torch.rand(M, K, device="cuda") * 2 - 1).bfloat16()
    w = (torch.rand(1, N, K, device="cuda") * 2 - 1).bfloat16()
    b = torch.randn(1, N, device="cuda")
    out = torch.empty(1, M, N, device="cuda", dtype=torch.bfloat16)
    res = {v: [] for v, _ in KERNELS}
    for r in range(ROUNDS + 1):
        for v, _ in KERNELS:
            ops.GEMM_VARIANT = v
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(3):
                ops.linear(x, w, N, prec, bias=b, out=out)
            e1.record()
            torch.cuda.synchronize()
            if r:
                res[v].append(e0.elapsed_time(e1) / 3)
    tiles = ((M + 255) // 256) * (N // 256)
    print(f"{name}: M={M} N={N} K={K}, {tiles} tiles = {tiles / 256:.2f} rounds of 256", flush=True)
    for v, vn in KERNELS:
        med = statistics.median(res[v])
        print(f"    {vn:32s} {med * 1e3:8.1f} us / launch   {med * 1e3 / (tiles / 256):6.2f} us / round   {2.0 * M * N * K / med / 1e9:6.0f} TF/s", flush=True)
ops.GEMM_VARIANT = None
