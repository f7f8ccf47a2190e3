"""Micro-benchmark of the mtt_gemm kernels on the real shapes of the NS-6 training step at per-GPU batch 63 (TFLOP/s per shape and
kernel).  Interleaved rounds inside ONE process (cdna_hip_programming.md §5.4 rule 24): every round runs every kernel once per shape;
the median and best round are printed.  Operands are uniform random (rule 25)."""
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
ops.call = lambda name, **kw: _call(name, **(dict(kw, variant=FORCE["v"]) if name == "gemm" else kw))     # mtt_gemm_desc.variant
prec = ops.Prec("bf16")
M63 = 63 * 1030
SHAPES = [("qkv", M63, 3072, 1024, 0), ("proj", M63, 1024, 1024, 0), ("proj+resid", M63, 1024, 1024, 2), ("fc1+gelu", M63, 4096, 1024, 1),
          ("fc2", M63, 1024, 4096, 0), ("fc2+resid", M63, 1024, 4096, 2), ("fc2 dgrad*gelu'", M63, 4096, 1024, 3), ("big", 8192, 8192, 8192, 0)]
KERNELS = [(20, "LDS-staged epilogue"), (19, "swapped MFMA + direct-store epilogue"), (13, "persistent, immediate stores")]
ROUNDS = int(sys.argv[1]) if len(sys.argv) > 1 else 5


def run(x, w, b, out, N, act):
    if act == 2:        # fp32 residual stream, in place, with per-sample DropPath scales (epilogue kind 3)
        ops.linear(x, w, N, prec, bias=b, out=XT, d_rows=(1030, 1030 * N, N), resid=XT, rowscale=RS, n_prompt=6, M=x.shape[0])
    elif act == 3:      # dgrad through GELU' (epilogue kind 4)
        ops.linear(x, w, N, prec, act=3, aux_in=AUX, out=out)
    else:
        ops.linear(x, w, N, prec, bias=b, act=act, out=out)


for name, M, N, K, act in SHAPES:
    x = (torch.rand(M, K, device="cuda") * 2 - 1).bfloat16()
    w = (torch.rand(1, N, K, device="cuda") * 2 - 1).bfloat16()
    b = torch.randn(1, N, device="cuda")
    out = torch.empty(1, M, ops.pad8(N), device="cuda", dtype=torch.bfloat16)
    XT = torch.zeros(M, N, device="cuda") if act == 2 else None
    RS = torch.ones(M // 1030 + 1, 2, device="cuda") if act == 2 else None
    AUX = torch.randn(1, M, N, device="cuda").bfloat16() if act == 3 else None
    res = {v: [] for v, _ in KERNELS}
    for v, _ in KERNELS:
        FORCE["v"] = v
        run(x, w, b, out, N, act)
    for _ in range(ROUNDS):
        for v, _ in KERNELS:
            FORCE["v"] = v
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(3):
                run(x, w, b, out, N, act)
            e1.record()
            torch.cuda.synchronize()
            res[v].append(e0.elapsed_time(e1) / 3)
    row = [f"{name:10s} M={M:6d} N={N:5d} K={K:5d}"]
    for v, vn in KERNELS:
        med, best = statistics.median(res[v]), min(res[v])
        row.append(f"{vn}: {2.0 * M * N * K / med / 1e9:6.0f} TF/s median ({2.0 * M * N * K / best / 1e9:6.0f} best, {med * 1e3:7.1f} us)")
    print("  |  ".join(row), flush=True)

# race screen: the persistent kernels accumulate in the same order as the one-tile-per-workgroup kernel and run the same epilogue
# arithmetic, so every launch must be BITWISE equal to it
M, N, K = M63, 3072, 1024
x = (torch.rand(M, K, device="cuda") * 2 - 1).bfloat16()
w = (torch.rand(1, N, K, device="cuda") * 2 - 1).bfloat16()
b = torch.randn(1, N, device="cuda")
ref = torch.empty(1, M, N, device="cuda", dtype=torch.bfloat16)
FORCE["v"] = 14
ops.linear(x, w, N, prec, bias=b, act=1, out=ref)
for v in (12, 13, 19):
    FORCE["v"] = v
    bad = 0
    for i in range(30):
        out = torch.full((1, M, N), 7.0, device="cuda", dtype=torch.bfloat16)
        ops.linear(x, w, N, prec, bias=b, act=1, out=out)
        bad += int(not torch.equal(out, ref))
    print(f"race screen variant {v}: " + ("clean (30 launches bitwise equal to the one-tile kernel)" if bad == 0 else f"{bad} mismatching launches"), flush=True)
