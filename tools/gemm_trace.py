"""Where a tile's time goes in the 256 x 256 LDS-DMA GEMM: per-workgroup timestamps from an EXPERIMENT build of the library
(`python multi-task-transformer_amd/_build.py --variant trace -DMTT_GEMM_TRACE=1 [-DMTT_RING=0]`): prologue / K loop / epilogue cycles per
workgroup and the gap between consecutive workgroups on the same CU (s_memrealtime, 10 ns ticks).  One launch per shape after a warm-up."""
import argparse
import ctypes
import os
import statistics
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--lib", default=os.path.join(ROOT, "build", "variants", "libmtt_trace.so"))
a = ap.parse_args()
import mtt_amd  # noqa: E402
mtt_amd._lib.LIB_PATH = os.path.abspath(a.lib)
from mtt_amd import ops  # noqa: E402

lib = mtt_amd._lib.load()
lib.mtt_debug_set_trace.argtypes = [ctypes.c_void_p]
lib.mtt_debug_set_trace.restype = None
prec = ops.Prec("bf16")
M63 = 63 * 1030
SHAPES = [("qkv", M63, 3072, 1024, 0), ("proj+resid", M63, 1024, 1024, 2), ("fc1+gelu", M63, 4096, 1024, 1), ("fc2+resid", M63, 1024, 4096, 2),
          ("big", 8192, 8192, 8192, 0)]

for name, M, N, K, act in SHAPES:
    x = (torch.rand(M, K, device="cuda") * 2 - 1).bfloat16()
    w = (torch.rand(1, N, K, device="cuda") * 2 - 1).bfloat16()
    b = torch.randn(1, N, device="cuda")
    out = torch.empty(1, M, ops.pad8(N), device="cuda", dtype=torch.bfloat16)
    XT = torch.zeros(M, N, device="cuda") if act == 2 else None
    RS = torch.ones(M // 1030 + 1, 2, device="cuda") if act == 2 else None

    def run():
        if act == 2:
            ops.linear(x, w, N, prec, bias=b, out=XT, d_rows=(1030, 1030 * N, N), resid=XT, rowscale=RS, n_prompt=6, M=x.shape[0])
        else:
            ops.linear(x, w, N, prec, bias=b, act=act, out=out)
    nwg = ((M + 255) // 256) * ((N + 255) // 256)
    buf = torch.zeros(nwg * 8, dtype=torch.int64, device="cuda")
    lib.mtt_debug_set_trace(None)
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    lib.mtt_debug_set_trace(ctypes.c_void_p(buf.data_ptr()))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    run()
    e1.record()
    torch.cuda.synchronize()
    lib.mtt_debug_set_trace(None)
    t = buf.cpu().view(nwg, 8).tolist()
    t = [r for r in t if r[4] > 0]
    pro = [r[2] - r[1] for r in t]
    loop = [r[3] - r[2] for r in t]
    epi = [r[4] - r[3] for r in t]
    tot = [r[4] - r[1] for r in t]
    rt = [r[5] - r[0] for r in t]                     # 10 ns ticks
    clk = statistics.mean(tot) / (statistics.mean(rt) * 10e-9) / 1e9 if statistics.mean(rt) > 0 else float("nan")
    by_cu = defaultdict(list)
    for r in t:
        key = ((r[6] >> 32) & 0xF, (r[6] >> 8) & 0xFF)      # (XCC_ID, HW_ID[15:8] = cu / sh / se)
        by_cu[key].append((r[0], r[5]))
    gaps, first = [], []
    t_min = min(r[0] for r in t)
    for k, v in by_cu.items():
        v.sort()
        first.append(v[0][0] - t_min)
        gaps += [v[i + 1][0] - v[i][1] for i in range(len(v) - 1)]
    us = lambda cyc: cyc / clk / 1e3
    med = statistics.median
    span = (max(r[5] for r in t) - t_min) * 0.01
    print(f"{name:11s} M={M} N={N} K={K}: {nwg} workgroups on {len(by_cu)} CUs, kernel {e0.elapsed_time(e1) * 1e3:.1f} us (trace span {span:.1f} us), shader clock ~{clk:.2f} GHz\n"
          f"    per workgroup (median cycles -> us): prologue {med(pro):7.0f} -> {us(med(pro)):5.2f}   K loop {med(loop):8.0f} -> {us(med(loop)):6.2f} "
          f"({us(med(loop)) / (K / 64):.3f} us per 64-deep K step)   epilogue {med(epi):7.0f} -> {us(med(epi)):5.2f}   total {us(med(tot)):6.2f}\n"
          f"    p10 / p90 us: prologue {us(sorted(pro)[len(pro) // 10]):.2f} / {us(sorted(pro)[len(pro) * 9 // 10]):.2f}, loop {us(sorted(loop)[len(loop) // 10]):.2f} / "
          f"{us(sorted(loop)[len(loop) * 9 // 10]):.2f}, epilogue {us(sorted(epi)[len(epi) // 10]):.2f} / {us(sorted(epi)[len(epi) * 9 // 10]):.2f}\n"
          f"    gap between consecutive workgroups on a CU: median {med(gaps) * 0.01 if gaps else float('nan'):.2f} us, mean {statistics.mean(gaps) * 0.01 if gaps else float('nan'):.2f} us; "
          f"first workgroup of a CU starts {med(first) * 0.01:.2f} us (median) / {max(first) * 0.01:.2f} us (max) after the first of the launch", flush=True)
