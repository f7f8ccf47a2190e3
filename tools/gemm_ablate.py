"""Where a K = 1024 tile of the 256 x 256 LDS-DMA GEMM spends its time: full kernel vs 'main loop only' (no epilogue) vs 'prologue +
epilogue only' (one K tile) on the NS-6 shapes — the ablation the guide prescribes before optimising (cdna_hip_programming.md §5.4).
Also a race screen: the balanced-schedule kernel must be BITWISE equal to the lock-step round-1 kernel (same accumulation order)."""
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import mtt_amd  # noqa: E402
from mtt_amd import ops  # noqa: E402

_call = ops.call
FORCE = {"v": 0}
ops.call = lambda name, **kw: _call(name, **(dict(kw, variant=FORCE["v"]) if name == "gemm" else kw))
prec = ops.Prec("bf16")
M63 = 63 * 1030
SHAPES = [("qkv", M63, 3072, 1024, 0, torch.bfloat16), ("proj(f32 out)", M63, 1024, 1024, 0, torch.float32), ("fc1+gelu", M63, 4096, 1024, 1, torch.bfloat16),
          ("fc2(f32 out)", M63, 1024, 4096, 0, torch.float32)]
KERNELS = [(3, "specialised epilogue"), (11, "general epilogue"), (6, "no epilogue")]
for name, M, N, K, act, odt in SHAPES:
    x = (torch.rand(M, K, device="cuda") * 2 - 1).bfloat16()
    w = (torch.rand(1, N, K, device="cuda") * 2 - 1).bfloat16()
    b = torch.randn(1, N, device="cuda")
    out = torch.empty(1, M, N, device="cuda", dtype=odt)
    res = {v: [] for v, _ in KERNELS}
    for _ in range(5):
        for v, _ in KERNELS:
            FORCE["v"] = v
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(3):
                ops.linear(x, w, N, prec, bias=b, act=act, out=out)
            e1.record()
            torch.cuda.synchronize()
            res[v].append(e0.elapsed_time(e1) / 3 * 1e3)
    rounds = -(-M // 256) * -(-N // 256) / 256.0
    print(f"{name:14s} M={M} N={N} K={K}: " + "  |  ".join(f"{n}: {statistics.median(res[v]):7.1f} us ({statistics.median(res[v]) / rounds:5.1f} us / tile round)" for v, n in KERNELS)
          + f"   [{rounds:.2f} rounds of 256 tiles]", flush=True)
# race screen
bad = 0
for name, M, N, K, act, odt in SHAPES[:3]:
    x = (torch.rand(M, K, device="cuda") * 2 - 1).bfloat16()
    w = (torch.rand(1, N, K, device="cuda") * 2 - 1).bfloat16()
    ref = torch.empty(1, M, N, device="cuda", dtype=torch.float32)
    FORCE["v"] = 4
    ops.linear(x, w, N, prec, out=ref)
    for v in (3, 5):
        for it in range(10):
            FORCE["v"] = v
            out = torch.full((1, M, N), float("nan"), device="cuda", dtype=torch.float32)
            ops.linear(x, w, N, prec, out=out)
            if not torch.equal(out, ref):
                bad += 1
                print(f"RACE SCREEN MISMATCH {name} variant {v} iter {it}: {int((out != ref).sum())} elements", flush=True)
print("race screen:", "clean (60 launches bitwise equal to the lock-step kernel)" if bad == 0 else f"{bad} mismatching launches")
