"""Micro-benchmark of the fp32-class flash forward on split planes (attn_fwd_x3_kernel), HIP-event timed, vs the fp64 emulator-free check
against the bf16 kernel's shape.  Usage: python tools/attn_x3_bench.py [B] [N] [nH] [T] [--lib build/variants/libmtt_<name>.so]"""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import mtt_amd  # noqa: E402
from mtt_amd import ops  # noqa: E402

argv = [a for a in sys.argv[1:]]
if "--lib" in argv:
    i = argv.index("--lib")
    mtt_amd._lib.LIB_PATH = os.path.abspath(argv[i + 1])
    del argv[i:i + 2]
print(f"library: {mtt_amd._lib.LIB_PATH}", flush=True)
B, N, nH, T = [int(a) for a in argv[:4]] + [63, 1030, 16, 6][len(argv):]
C = nH * 64
x3f = ops.Prec("x3f")
dev = torch.device("cuda")
q = torch.randn(B * N, 3 * C, device=dev)
qh = q.to(torch.bfloat16)
qs = ops.Split(qh, (q - qh.float()).to(torch.bfloat16))


def timed(fn, iters=10):
    for _ in range(3):
        fn()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / iters * 1e3)
    return statistics.median(ts)


t3 = timed(lambda: ops.attention(qs, B, N, nH, T, x3f, want_lse=True))
tb = timed(lambda: ops.attention(qh, B, N, nH, T, ops.Prec("bf16"), want_lse=True))
o3, raw3, lse3 = ops.attention(qs, B, N, nH, T, x3f, want_lse=True)
# reference: fp64 softmax attention of the first image / two heads on the summed planes
xq = (qs.hi[:N].double() + qs.lo[:N].double()).view(N, 3, nH, 64)
errs = []
for h in (0, nH - 1):
    S = xq[:, 0, h] @ xq[:, 1, h].t()
    P = torch.softmax(S * 0.125, -1)
    O = P @ xq[:, 2, h]
    got = (o3.hi[:N].double() + o3.lo[:N].double()).view(N, nH, 64)[:, h]
    errs.append(float((got - O).norm() / O.norm()))
    if T:
        errs.append(float((raw3[0, h].double() - S[:T]).norm() / S[:T].norm()))
gf = 4.0 * N * N * 64 * nH * B / 1e9
print(f"attention forward B={B} N={N} nH={nH} T={T}: x3 on planes {t3:.0f} us ({3 * gf / t3 * 1e3:.0f} TFLOP/s of MFMA work), bf16 {tb:.0f} us "
      f"({gf / tb * 1e3:.0f} TFLOP/s); x3 vs fp64 (image 0, heads 0 / {nH - 1}: out, rawlog): {['%.1e' % e for e in errs]}", flush=True)
