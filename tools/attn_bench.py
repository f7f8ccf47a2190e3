"""Micro-benchmark of the attention kernels through the C ABI (HIP-event timed).  Usage: python tools/attn_bench.py [B] [N] [nH] [T]
MTT_ATTN_PLAIN=1 forces the straightforward forward kernel (mtt_attn_desc.variant)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import mtt_amd  # noqa: E402
from mtt_amd import ops  # noqa: E402

B, N, nH, T = [int(a) for a in sys.argv[1:5]] + [40, 1030, 16, 6][len(sys.argv) - 1:]
C = nH * 64
prec = ops.Prec("bf16")
_call = ops.call
FORCE = {"v": 1 if os.environ.get("MTT_ATTN_PLAIN") == "1" else 0}
ops.call = lambda name, **kw: _call(name, **(dict(kw, variant=FORCE["v"]) if name == "attn_fwd" else kw))
dev = torch.device("cuda")
qkv = (torch.randn(B * N, 3 * C, device=dev) * 1.0).to(torch.bfloat16)
dao = torch.randn(B * N, C, device=dev).to(torch.bfloat16)
drawlog = torch.randn(B, nH, T, N, device=dev) * 0.01 if T else None


def timed(fn, iters=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


ao, rawlog, lse = ops.attention(qkv, B, N, nH, T, prec, want_lse=True)
t_f = timed(lambda: ops.attention(qkv, B, N, nH, T, prec, want_lse=True))
if FORCE["v"] == 0:                                    # A/B of the forward variants (mtt_attn_desc.variant), interleaved, + bitwise comparison
    import statistics
    VARS = (0, 2, 3)  # 0 = default, 2 = MTT_ATTN_FAST_V0 (the first forward kernel), 3 = MTT_ATTN_FAST_V1 (register-staged tiles)
    res = {v: [] for v in VARS}
    for _ in range(5):
        for v in VARS:
            FORCE["v"] = v
            res[v].append(timed(lambda: ops.attention(qkv, B, N, nH, T, prec, want_lse=True), 5))
    eq = {}
    for v in VARS[1:]:
        FORCE["v"] = v
        ao2, rawlog2, lse2 = ops.attention(qkv, B, N, nH, T, prec, want_lse=True)
        eq[v] = bool(torch.equal(ao, ao2) and torch.equal(lse, lse2) and (not T or torch.equal(rawlog, rawlog2)))
    FORCE["v"] = 0
    print("forward A/B (us, median of 5): " + ", ".join(f"variant {v}: {statistics.median(res[v]) * 1e3:.0f}" for v in VARS) + f"; bitwise equal to variant 0: {eq}")
dqkv = torch.empty_like(qkv)
dsum = torch.empty(B, nH, 2, (N + 3) // 4 * 4, device=dev)


BV = {"v": 0}


def bwd():
    _call("attn_bwd", qkv=qkv, out=ao, rawlog=None, lse=lse, B=B, N=N, nH=nH, T=T, dtype=1, prec=0, scale=0.125, variant=BV["v"],
          xargs=[dao, drawlog, dqkv, dsum])


t_b = timed(bwd)
import statistics as _st
rb = {0: [], 3: [], 2: []}
for _ in range(5):
    for v in rb:
        BV["v"] = v
        rb[v].append(timed(bwd, 5))
BV["v"] = 0
bwd(); ref = dqkv.clone()
same = {}
for v in (3, 2):
    BV["v"] = v
    bwd(); same[v] = torch.equal(ref, dqkv)
BV["v"] = 0
print(f"backward A/B (us, median of 5): default (LDS-DMA) {_st.median(rb[0]) * 1e3:.0f}, MTT_ATTN_FAST_V1 {_st.median(rb[3]) * 1e3:.0f}, MTT_ATTN_FAST_V0 {_st.median(rb[2]) * 1e3:.0f}; dqkv bitwise equal to default: {same}")
gf = 4.0 * N * N * 64 * nH * B / 1e9
print(f"attention B={B} N={N} nH={nH} T={T} plain={os.environ.get('MTT_ATTN_PLAIN', '0')}: fwd {t_f * 1e3:.0f} us = {gf / t_f:.0f} TFLOP/s (2 GEMMs);"
      f"  bwd {t_b * 1e3:.0f} us = {2.5 * gf / t_b:.0f} TFLOP/s (5 GEMMs algorithmic)")
