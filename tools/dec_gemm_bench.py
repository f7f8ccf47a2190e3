"""Task-batched decoder GEMMs of NS-6 at per-GPU batch 63 (32 x 32 maps, M = 64 512 rows per task): forward 1x1s and their input
gradients on the 128-wide register-staged kernel (variant 1) against the 256 x 256 LDS-DMA kernel (variant 3).  Outputs of 300 / 350
channels waste 41 / 32 % of a 256-wide tile pair; does the faster kernel still win?"""
import importlib
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import mtt_amd  # noqa: E402,F401
from mtt_amd import ops  # noqa: E402

ap = importlib.import_module("multi-task-transformer_amd.autograd_path")
prec = ops.Prec("bf16")
M = 63 * 1024
M_OF = {"head linear_pred (128 x 128 map, 21 classes)": 63 * 128 * 128, "channel attention, prompt rows (M = 378)": 63 * 6}


def timed(fn, rounds=5, inner=4):
    fn()
    fn()
    ts = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(inner):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / inner)
    return statistics.median(ts)


# (name, Z, N, K) : D[z] [M, pad8(N)] = A[z] [M, pad8(K)] @ W[z] [N, pad8(K)]^T + bias
for name, Z, N, K in (("fea_decode fwd (12 = 6 tasks x spa/chan)", 12, 300, 1024), ("fea_fuse[0] fwd", 6, 350, 608), ("fea_fuse[4] fwd", 6, 350, 350),
                      ("fea_decode dgrad", 12, 1024, 300), ("fea_fuse[0] dgrad", 6, 608, 350), ("fea_fuse[4] dgrad", 6, 350, 350),
                      ("head linear_pred (128 x 128 map, 21 classes)", 1, 21, 350), ("encoder qkv (for reference)", 1, 3072, 1024),
                      ("channel attention, prompt rows (M = 378)", 1, 1024, 1024)):
    M = M_OF.get(name, 63 * 1024)
    Np, Kp = ops.pad8(N), ops.pad8(K)
    A = (torch.rand(Z, M, Kp, device="cuda") - 0.5).bfloat16()
    A[..., K:] = 0
    W = (torch.rand(Z, N, Kp, device="cuda") - 0.5).bfloat16()
    W[..., K:] = 0
    bias = torch.randn(Z, N, device="cuda")
    D = torch.empty(Z, M, Np, dtype=torch.bfloat16, device="cuda")
    fl = 2.0 * Z * M * N * K
    ref = None
    for variant, label in ((1, "register-staged 128 x 128"), (3, "LDS-DMA 256 x 256"), (4, "LDS-DMA 128 x 128, 2 WG/CU")):
        def go():
            ap._gemm(A, W, D, M, N, Kp, prec, lda=Kp, ldb=Kp, ldd=Np, batch=Z, a_zo=M * Kp, b_zo=N * Kp, d_zo=M * Np, colshift=bias, col_zo=N,
                     n_store=Np, variant=variant)
        t = timed(go)
        out = D.float().clone()
        if ref is None:
            ref = out
        print(f"{name:42s} Z={Z:2d} N={N:4d} K={K:4d}  {label:26s}: {t * 1e3:8.1f} us  {fl / t / 1e9:6.0f} TF/s   rel diff {float((out - ref).norm() / ref.norm()):.1e}",
              flush=True)
