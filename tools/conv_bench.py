"""Micro-benchmark of the implicit-GEMM 3x3 conv (ConvHead shape: 128x128 maps, 350 -> 350 channels, 6 tasks) through the C ABI."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import mtt_amd  # noqa: E402
from mtt_amd import ops  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
H = W = int(sys.argv[2]) if len(sys.argv) > 2 else 128
Z, C = 6, 350
prec = ops.Prec("bf16")
Cp = ops.pad8(C)
x = torch.randn(Z, B * H * W, Cp, device="cuda").bfloat16()
x[..., C:] = 0
ws = [torch.randn(C, C, 3, 3, device="cuda") * 0.02 for _ in range(Z)]
wp = ops.pack_conv3(ws, prec, "bench")
wd = ops.pack_conv3(ws, prec, "bench", transpose=True)


def timed(fn, iters=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


fl = 2.0 * B * H * W * C * C * 9 * Z
t = timed(lambda: ops.conv3x3(x, wp, C, C, B, H, W, prec))
print(f"conv3x3 fwd   B={B} {H}x{W} C={C} Z={Z}: {t:.3f} ms = {fl / t / 1e9:.0f} TFLOP/s")
t = timed(lambda: ops.conv3x3(x, wd, C, C, B, H, W, prec, flip=1))
print(f"conv3x3 dgrad B={B} {H}x{W} C={C} Z={Z}: {t:.3f} ms = {fl / t / 1e9:.0f} TFLOP/s")
