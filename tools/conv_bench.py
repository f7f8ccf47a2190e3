"""Micro-benchmark of the implicit-GEMM 3x3 conv (ConvHead shape: 128x128 maps, 350 -> 350 channels, 6 tasks; fea_fuse shape: 32x32)
through the C ABI: register-staged 128x128 kernel vs the phased LDS-DMA kernel, interleaved rounds in one process."""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import mtt_amd  # noqa: E402
from mtt_amd import ops  # noqa: E402

_call = ops.call
FORCE = {"v": 0}
ops.call = lambda name, **kw: _call(name, **(dict(kw, variant=FORCE["v"]) if name == "gemm" else kw))
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
Z, C = 6, 350
prec = ops.Prec("bf16")
Cp = ops.pad8(C)
ws = [torch.randn(C, C, 3, 3, device="cuda") * 0.02 for _ in range(Z)]
wp = ops.pack_conv3(ws, prec, "bench")
wd = ops.pack_conv3(ws, prec, "bench", transpose=True)
for H, Bx in ((128, B), (32, B * 4)):
    W = H
    x = torch.randn(Z, Bx * H * W, Cp, device="cuda").bfloat16()
    x[..., C:] = 0
    fl = 2.0 * Bx * H * W * C * C * 9 * Z
    for tag, pack, flip in (("fwd", wp, 0), ("dgrad", wd, 1)):
        res = {1: [], 3: []}
        for v in res:
            FORCE["v"] = v
            ops.conv3x3(x, pack, C, C, Bx, H, W, prec, flip=flip)
        for _ in range(4):
            for v in res:
                FORCE["v"] = v
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize()
                e0.record()
                for _ in range(2):
                    ops.conv3x3(x, pack, C, C, Bx, H, W, prec, flip=flip)
                e1.record()
                torch.cuda.synchronize()
                res[v].append(e0.elapsed_time(e1) / 2)
        print(f"conv3x3 {tag:5s} B={Bx} {H}x{W} C={C} Z={Z}: " + "  |  ".join(
            f"{n}: {fl / statistics.median(res[v]) / 1e9:6.0f} TF/s ({statistics.median(res[v]):.3f} ms)" for v, n in ((1, "reg128"), (3, "dma phased"))), flush=True)
