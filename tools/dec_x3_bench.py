"""The decoder's task-batched 1x1 GEMMs of the fp32-class forward (x3f) at per-GPU batch 63: the register-staged x3 kernel on fp32 operands
(gemm_kernel<K, K, 2>: splits while staging) against the split-plane LDS-DMA kernel (gemm_ring3_kernel) on pre-split planes, + the
split_cast pass that makes planes out of an fp32 activation.  Shapes: fea_decode (12 x [64512, 1024] -> 300, written as 6 padded pairs),
fea_fuse[0] (6 x K = 608 -> 350), fea_fuse[4] (6 x K = 352 -> 350)."""
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import mtt_amd  # noqa: E402
from mtt_amd import ops  # noqa: E402

if "--lib" in sys.argv:                      # A/B builds (python multi-task-transformer_amd/_build.py --variant NAME -D...): one library per process
    mtt_amd._lib.LIB_PATH = os.path.abspath(sys.argv[sys.argv.index("--lib") + 1])
print(f"library: {mtt_amd._lib.LIB_PATH}", flush=True)

x3, x3f = ops.Prec("x3"), ops.Prec("x3f")
M = 63 * 1024


def timed(fn, rounds=3):
    fn()
    ts = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(2):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 2 * 1e3)
    return statistics.median(ts)


for name, Z, N, K, pair in (("fea_decode", 12, 300, 1024, True), ("fea_fuse0", 6, 350, 608, False), ("fea_fuse4", 6, 350, 352, False)):
    Np = ops.pad8(N)
    x = torch.randn(Z, M, K, device="cuda")
    ws = [torch.nn.Parameter(torch.randn(N, K, device="cuda") * 0.05) for _ in range(Z)]
    b = torch.randn(Z, N, device="cuda")
    w32 = ops.pack_linear(ws, x3, ("b32", name))
    wsp = ops.pack_linear_split(ws, ("bsp", name))
    kw = dict(batch_inner=2, d_z=(M * 2 * Np, Np), ldd=2 * Np, n_store=Np) if pair else {}
    shape = (Z // 2, M, 2 * Np) if pair else (Z, M, Np)
    o32 = torch.empty(shape, device="cuda")
    osp = ops.Split.empty(shape, "cuda")
    t_reg = timed(lambda: ops.linear(x, w32, N, x3, bias=b, out=o32, **kw))
    xs = ops.Split.empty((Z, M, K), "cuda")
    t_cast = timed(lambda: ops.call("split_cast", args=[x.view(Z * M, K), xs.hi.view(Z * M, K), xs.lo.view(Z * M, K), Z * M, K, K, K]))
    o2 = torch.empty(shape, device="cuda")
    t_ring = timed(lambda: ops.linear(xs, wsp, N, x3f, bias=b, out=o2, **kw))
    t_ring_sp = timed(lambda: ops.linear(xs, wsp, N, x3f, bias=b, out=osp, **kw))
    err = float((o2 - o32).norm() / o32.norm())
    err_sp = float((osp.hi.float() + osp.lo.float() - o32).norm() / o32.norm())
    print(f"{name:10s} Z={Z} M={M} N={N} K={K}: register-staged x3 {t_reg:7.1f} us | ring3 on planes {t_ring:7.1f} us (fp32 out; split out {t_ring_sp:7.1f} us), "
          f"split_cast of A {t_cast:6.1f} us | rel diff {err:.1e} / {err_sp:.1e}", flush=True)

# the fea_fuse 3x3 conv (6 x [63 * 32 * 32, 352] -> 350, K = 9 * 352) and the nine-tap head GEMM (N = 9 * 352) on planes
Z, Ci, Co, B, H, W = 6, 350, 350, 63, 32, 32
Cp = ops.pad8(Ci)
xs = ops.Split.empty((Z, B * H * W, Cp), "cuda")
xs.hi.normal_(); xs.lo.normal_(std=1e-3)
ws = [torch.nn.Parameter(torch.randn(Co, Ci, 3, 3, device="cuda") * 0.05) for _ in range(Z)]
wc = ops.pack_conv3_split(ws, "bconv")
t_conv = timed(lambda: ops.conv3x3(xs, wc, Co, Ci, B, H, W, x3f, out_dtype=torch.float32))
w9 = ops.pack_upconv9_split(ws, "b9")
t_nine = timed(lambda: ops.linear(xs, w9, w9.shape[1], x3f, out_dtype=torch.float32))
print(f"conv3x3 on planes (N = 350, K = 3168, z = 6): {t_conv:7.1f} us | nine-tap head GEMM (N = {w9.shape[1]}, K = 352, z = 6): {t_nine:7.1f} us", flush=True)
