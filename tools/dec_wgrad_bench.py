"""Task-batched weight gradients of the decoder at the NS-6 shapes (per-GPU batch 63, 32 x 32 maps): the 128-wide register-staged
kernel (what the policy picks for N = 300 / 350 outputs today) against the token-major LDS-DMA kernel (variant 3 forces it) over a
sweep of reduction slices.  dW[z] = dy[z]^T x[z], fp32 slabs summed afterwards."""
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


def timed(fn, rounds=5):
    fn()
    ts = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return statistics.median(ts)


def run(dy, x, N, Kp, Z, S, lda, ldb, a_zo, b_zo, variant):
    c = (M // S) // 64 * 64
    nz = M // c
    rem = M - nz * c
    slabs = torch.empty(Z, nz + (1 if rem else 0), N, Kp, dtype=torch.float32, device="cuda")
    SS = slabs.shape[1]

    def go():
        ap._gemm(dy, x, slabs, N, Kp, c, prec, a_op=ap.OP_R, b_op=ap.OP_R, lda=lda, ldb=ldb, ldd=Kp, batch=Z * nz, batch_inner=nz,
                 a_zo=a_zo, a_zi=c * lda, b_zo=b_zo, b_zi=c * ldb, d_zo=SS * N * Kp, d_zi=N * Kp, variant=variant)
        if rem:
            ap._gemm(dy.reshape(-1)[nz * c * lda:], x.reshape(-1)[nz * c * ldb:], slabs[:, nz], N, Kp, rem, prec, a_op=ap.OP_R, b_op=ap.OP_R,
                     lda=lda, ldb=ldb, ldd=Kp, batch=Z, a_zo=a_zo, b_zo=b_zo, d_zo=SS * N * Kp, variant=variant)
        return slabs.sum(1)
    return go


for name, N, K, Z, pair in (("fea_decode (one half of the pair)", 300, 1024, 6, True), ("fea_fuse[0]", 350, 608, 6, False),
                            ("fea_fuse[4]", 350, 352, 6, False)):
    Np, Kp = ops.pad8(N), ops.pad8(K)
    lda = 2 * Np if pair else Np
    dy = (torch.rand(Z, M, lda, device="cuda") - 0.5).bfloat16()
    x = (torch.rand((2 * Z if pair else Z), M, Kp, device="cuda") - 0.5).bfloat16()
    a_zo, b_zo = M * lda, (2 if pair else 1) * M * Kp
    fl = 2.0 * Z * M * N * K
    ref = None
    for variant, label, sweep in ((1, "register-staged 128 x 128", (8, 12, 16)), (3, "token-major LDS-DMA 256 x 256", (2, 3, 4, 5, 6, 8, 10, 12, 16))):
        for S in sweep:
            go = run(dy, x, N, Kp, Z, S, lda, Kp, a_zo, b_zo, variant)
            t = timed(go)
            out = go()
            if ref is None:
                ref = out
            err = float((out - ref).norm() / ref.norm())
            print(f"{name:36s} N={N} K={K} Z={Z}  {label:32s} S={S:2d}: {t * 1e3:8.1f} us  {fl / t / 1e9:6.0f} TF/s   rel diff vs first {err:.1e}", flush=True)
