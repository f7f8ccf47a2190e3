"""ConvHead's first stage (x4 bilinear resize + task-batched 3x3 conv, NS-6 shape: 6 tasks, 32x32 -> 128x128, 350 channels) both ways
through the autograd Functions: the reference's operation order (BilinearFn + Conv3x3Fn on the upsampled stack) against the taps-first
form (UpConv3x3Fn: one GEMM on the h x w map + mtt_upconv4_expand / _gather).  Prints forward / backward times and per-kernel
pieces of the fused path."""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import mtt_amd  # noqa: E402
from mtt_amd import autograd_path as ap  # noqa: E402
from mtt_amd import ops  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 63
Z, C, h, w = 6, 350, 32, 32
prec = ops.Prec("bf16")
Cp = ops.pad8(C)
torch.manual_seed(0)
ws = [torch.nn.Parameter(torch.randn(C, C, 3, 3, device="cuda") * 0.02) for _ in range(Z)]
bs = [torch.nn.Parameter(torch.randn(C, device="cuda") * 0.1) for _ in range(Z)]
x = torch.zeros(Z, B * h * w, Cp, device="cuda")
x[..., :C] = torch.randn(Z, B * h * w, C, device="cuda")
x.requires_grad_(True)
dy = torch.zeros(Z, B * 16 * h * w, Cp, device="cuda", dtype=torch.bfloat16)
dy[..., :C] = torch.randn(Z, B * 16 * h * w, C, device="cuda").bfloat16() * 0.01


def unfused():
    up = ap.upsample4(x, B, h, w, prec)
    return ap.Conv3x3Fn.apply(up, (B, 4 * h, 4 * w, C, C), prec, "hc", *ws, *bs)


def fused():
    return ap.UpConv3x3Fn.apply(x, (B, h, w, C, C), prec, "hc9", *ws, *bs)


def timed(fn, n=3):
    ts = []
    for _ in range(n + 1):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        r = fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return statistics.median(ts[1:]), r


res = {}
for name, fn in (("unfused", unfused), ("fused", fused)):
    t_f, y = timed(fn)
    grads = {}

    def bwd():
        x.grad = None
        for q in ws + bs:
            q.grad = None
        yy = fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        yy.backward(dy)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1)
    tb = statistics.median([bwd() for _ in range(4)][1:])
    res[name] = (y.float(), x.grad.clone(), ws[0].grad.clone(), bs[0].grad.clone())
    print(f"{name:8s} B={B}: forward {t_f:7.2f} ms   backward {tb:7.2f} ms   (peak mem {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB)", flush=True)
    del y
    torch.cuda.empty_cache()
for i, what in enumerate(("y", "dx", "dW[0]", "db[0]")):
    a, b = res["fused"][i], res["unfused"][i]
    print(f"  fused vs unfused {what:6s}: rel diff {float((a - b).norm() / b.norm()):.2e}")

# pieces of the fused path
xa = x.detach().bfloat16()
w9 = ops.pack_upconv9(ws, prec, "hc9")
bias = torch.stack([b.detach() for b in bs])
t, z = timed(lambda: ops.linear(xa, w9, w9.shape[1], prec))
print(f"  tap GEMM [{B * h * w} x {w9.shape[1]} x {Cp}] x {Z}: {t:.3f} ms = {2 * B * h * w * w9.shape[1] * Cp * Z / t / 1e9:.0f} TFLOP/s")
t, y = timed(lambda: ops.upconv4_expand(z, C, B, h, w, bias=bias))
print(f"  upconv4_expand: {t:.3f} ms = {(z.numel() + y.numel()) * 2 / t / 1e6:.0f} GB/s of algorithmic traffic (read z + write y)")
t, dz = timed(lambda: ops.upconv4_gather(dy, C, B, h, w))
print(f"  upconv4_gather: {t:.3f} ms = {(dz.numel() + dy.numel()) * 2 / t / 1e6:.0f} GB/s of algorithmic traffic (read dy + write dz)")
