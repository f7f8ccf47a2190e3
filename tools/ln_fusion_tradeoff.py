"""north_star's "LayerNorm + prompt-add fusion", measured on THIS library: what a LayerNorm folded into the GEMM
prologue would save against what it costs.  A prologue fusion needs the A operand in REGISTERS (x is fp32 in HBM; (x - mean) * rstd * g + b
is applied while staging), i.e. the register-staged kernel instead of the LDS-DMA kernel.  Lower bound of the fused variant's time = the
register-staged kernel reading fp32 A and merely ROUNDING it while staging (gemm_kernel<K, K, 1>: no LayerNorm math at all, no statistics
pass) — if even that is slower than `ln_fwd + LDS-DMA GEMM`, the fusion cannot win.  Shapes: the qkv / fc1 call sites at per-GPU batch 63."""
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import mtt_amd  # noqa: E402
from mtt_amd import ops  # noqa: E402

prec = ops.Prec("bf16")
M = 63 * 1030


def timed(fn, rounds=5):
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
        ts.append(e0.elapsed_time(e1) / 3 * 1e3)
    return statistics.median(ts)


for name, N, K, act in (("qkv", 3072, 1024, 0), ("fc1+gelu", 4096, 1024, 1)):
    x32 = torch.randn(M, K, device="cuda")
    g, b = torch.randn(K, device="cuda"), torch.randn(K, device="cuda")
    w = (torch.rand(1, N, K, device="cuda") * 2 - 1).bfloat16()
    bias = torch.randn(1, N, device="cuda")
    out = torch.empty(1, M, N, device="cuda", dtype=torch.bfloat16)
    xn = ops.layernorm(x32, g, b, 1e-6, prec)[0]
    t_ln = timed(lambda: ops.layernorm(x32, g, b, 1e-6, prec))
    t_dma = timed(lambda: ops.linear(xn, w, N, prec, bias=bias, act=act, out=out))
    kw = dict(A=x32, B=w, D=out, M=M, N=N, K=K, a_op=0, b_op=0, a_dtype=0, b_dtype=1, d_dtype=1, prec=0, lda=K, ldb=K, ldd=N, batch=1,
              batch_inner=1, alpha=1.0, colshift=bias, n_store=N, act=act)
    assert mtt_amd._lib.gemm_variant(**kw) == 0                       # fp32 A: only the register-staged kernel takes it
    t_reg = timed(lambda: ops.call("gemm", **kw))
    kw16 = dict(kw, A=xn, a_dtype=1, variant=1)                       # the same kernel on bf16 A, for reference
    t_reg16 = timed(lambda: ops.call("gemm", **kw16))
    print(f"{name:9s} M={M} N={N} K={K}:  ln_fwd {t_ln:6.1f} us + LDS-DMA GEMM {t_dma:6.1f} us = {t_ln + t_dma:6.1f} us   |   register-staged GEMM, fp32 A rounded while "
          f"staging (lower bound of a LayerNorm-prologue kernel) {t_reg:6.1f} us   |   register-staged GEMM on bf16 A {t_reg16:6.1f} us   ->  fusion would "
          f"{'save at most' if t_reg < t_ln + t_dma else 'LOSE at least'} {abs(t_ln + t_dma - t_reg):.1f} us per call", flush=True)
