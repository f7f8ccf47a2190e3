"""Micro-benchmark of the LDS-DMA GEMM kernel on the real shapes of the NS-6 training step at per-GPU batch 63 (TFLOP/s per shape).
`--lib PATH` benches another build of the library (e.g. one compiled with -DMTT_GROUP_M=8) — one library per process; median and best
of ROUNDS rounds of 3 launches.  Operands are uniform random (cdna_hip_programming.md §5.4 rule 25)."""
import argparse
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--lib", default=None)
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--variant", type=int, default=None, help="force mtt_gemm_desc.variant (3 = 256 x 256 LDS-DMA, 4 = 128 x 128 LDS-DMA, 1 = register-staged)")
ap.add_argument("--no-check", action="store_true", help="skip the correctness / race screens (A/B timing of variant builds)")
ap.add_argument("--split", action="store_true", help="also the x3 product on MTT_SPLIT planes (gemm_dma_kernel<2>)")
a = ap.parse_args()
import mtt_amd  # noqa: E402
if a.lib:
    mtt_amd._lib.LIB_PATH = os.path.abspath(a.lib)
from mtt_amd import ops  # noqa: E402

prec = ops.Prec("bf16")
ops.GEMM_VARIANT = a.variant
M63 = 63 * 1030
SHAPES = [("qkv", M63, 3072, 1024, 0), ("proj+resid", M63, 1024, 1024, 2), ("fc1+gelu", M63, 4096, 1024, 1),
          ("fc2+resid", M63, 1024, 4096, 2), ("fc2 dgrad*gelu'", M63, 4096, 1024, 3), ("big", 8192, 8192, 8192, 0)]


def run(x, w, b, out, N, act):
    if act == 2:        # fp32 residual stream, in place, with per-sample DropPath scales (epilogue kind 3)
        ops.linear(x, w, N, prec, bias=b, out=XT, d_rows=(1030, 1030 * N, N), resid=XT, rowscale=RS, n_prompt=6, M=x.shape[0])
    elif act == 3:      # dgrad through GELU' (epilogue kind 4)
        ops.linear(x, w, N, prec, act=3, aux_in=AUX, out=out)
    else:
        ops.linear(x, w, N, prec, bias=b, act=act, out=out)


def timed(fn, rounds):
    fn()
    ts = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(3):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 3)
    return statistics.median(ts), min(ts)


print(f"library: {mtt_amd._lib.LIB_PATH}", flush=True)
tot = 0.0
for name, M, N, K, act in SHAPES:
    x = (torch.rand(M, K, device="cuda") * 2 - 1).bfloat16()
    w = (torch.rand(1, N, K, device="cuda") * 2 - 1).bfloat16()
    b = torch.randn(1, N, device="cuda")
    out = torch.empty(1, M, ops.pad8(N), device="cuda", dtype=torch.bfloat16)
    XT = torch.zeros(M, N, device="cuda") if act == 2 else None
    RS = torch.ones(M // 1030 + 1, 2, device="cuda") if act == 2 else None
    AUX = torch.randn(1, M, N, device="cuda").bfloat16() if act == 3 else None
    med, best = timed(lambda: run(x, w, b, out, N, act), a.rounds)
    tot += med if name != "big" else 0.0
    line = f"{name:16s} M={M:6d} N={N:5d} K={K:5d}: {2.0 * M * N * K / med / 1e9:6.0f} TF/s median ({2.0 * M * N * K / best / 1e9:6.0f} best, {med * 1e3:7.1f} us)"
    if a.split and name != "big":
        x3 = ops.Prec("x3f")
        xs = ops.split_cast((torch.rand(M, K, device="cuda") * 2 - 1))
        wp = torch.nn.Parameter(torch.rand(N, K, device="cuda") * 2 - 1)
        ws = ops.pack_linear_split([wp], ("bench", name))
        o32 = torch.empty(1, M, N, device="cuda")
        m2, b2 = timed(lambda: ops.linear(xs, ws, N, x3, bias=b, out=o32), a.rounds)
        line += f"  |  split x3: {2.0 * M * N * K / m2 / 1e9:6.0f} TF/s effective = {6.0 * M * N * K / m2 / 1e9:6.0f} MFMA TF/s ({m2 * 1e3:7.1f} us)"
    print(line, flush=True)
print(f"sum of the step's five shapes: {tot * 1e3:.1f} us", flush=True)

if a.no_check:
    sys.exit(0)

# correctness screen of the loaded build: the LDS-DMA kernel against the register-staged general kernel (ragged M; K from the shortest legal
# reduction — 64 = two 32-deep ring steps, shorter than the ring — to 4096), 20 launches each bitwise equal to the first (race screen)
for (M, N, K) in ((M63, 1024, 1024), (5000, 768, 4096), (777, 512, 64), (3000, 1280, 128), (2049, 512, 192), (1500, 768, 320)):
    x = (torch.rand(M, K, device="cuda") * 2 - 1).bfloat16()
    w = (torch.rand(1, N, K, device="cuda") * 2 - 1).bfloat16()
    b = torch.randn(1, N, device="cuda")
    ref = torch.empty(1, M, N, device="cuda")
    ops.call("gemm", A=x, B=w, D=ref, M=M, N=N, K=K, a_op=0, b_op=0, a_dtype=1, b_dtype=1, d_dtype=0, prec=0, lda=K, ldb=K, ldd=N, batch=1,
             batch_inner=1, alpha=1.0, colshift=b, n_store=N, variant=1)
    first, bad = None, 0
    for i in range(20):
        out = torch.full((1, M, N), 7.0, device="cuda")
        ops.call("gemm", A=x, B=w, D=out, M=M, N=N, K=K, a_op=0, b_op=0, a_dtype=1, b_dtype=1, d_dtype=0, prec=0, lda=K, ldb=K, ldd=N, batch=1,
                 batch_inner=1, alpha=1.0, colshift=b, n_store=N, variant=a.variant or 3)
        first = out if first is None else first
        bad += int(not torch.equal(out, first))
    err = float((first - ref).norm() / ref.norm())
    print(f"check M={M} N={N} K={K}: rel diff vs general kernel {err:.2e}, {bad} of 20 launches differ from the first", flush=True)

# the split-plane (x3) kernel against an fp64 product of the SAME split operands (hi + lo), same K sweep + race screen
x3 = ops.Prec("x3f")
for (M, N, K) in ((M63, 1024, 1024), (5000, 768, 4096), (777, 512, 64), (3000, 1280, 128), (2049, 512, 192)):
    xs = ops.split_cast((torch.rand(M, K, device="cuda") * 2 - 1))
    wp = torch.nn.Parameter(torch.rand(N, K, device="cuda") * 2 - 1)
    ws = ops.pack_linear_split([wp], ("chk", M, N, K))
    b = torch.randn(1, N, device="cuda")
    ref = (xs.hi.double() + xs.lo.double()) @ (ws.hi[0].double() + ws.lo[0].double()).t() + b.double()
    first, bad = None, 0
    for i in range(10):
        o32 = torch.full((1, M, N), 7.0, device="cuda")
        ops.linear(xs, ws, N, x3, bias=b, out=o32)
        first = o32 if first is None else first
        bad += int(not torch.equal(o32, first))
    err = float((first[0].double() - ref).norm() / ref.norm())
    print(f"check split x3 M={M} N={N} K={K}: rel diff vs fp64 product of the planes {err:.2e}, {bad} of 10 launches differ from the first", flush=True)
