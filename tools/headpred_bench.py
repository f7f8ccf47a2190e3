"""The heads' 1x1 predictions in the fp32-class modes (M = B 128 128 rows, K = 352, N = 1 .. 21, fp32 operands): the exact-fp32 MFMA kernel
(mtt_gemm variant 11, gemm_f32n_kernel) against the register-staged x3 kernel (variant = MTT_GEMM_GENERAL), same process."""
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import mtt_amd  # noqa: E402
from mtt_amd import ops  # noqa: E402

M, K = 63 * 128 * 128, 352


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


a = torch.randn(M, K, device="cuda")
for N in (21, 7, 1):
    w = torch.randn(N, K, device="cuda") * 0.1
    b = torch.randn(N, device="cuda")
    Np = ops.pad8(N)
    kw = dict(A=a, B=w, M=M, N=N, K=K, a_op=0, b_op=0, a_dtype=0, b_dtype=0, d_dtype=0, prec=1, lda=K, ldb=K, ldd=Np, batch=1, batch_inner=1, alpha=1.0,
              colshift=b, n_store=Np)
    o1, o2 = torch.empty(M, Np, device="cuda"), torch.empty(M, Np, device="cuda")
    v = mtt_amd._lib.gemm_variant(D=o1, **kw)
    t_new = timed(lambda: ops.call("gemm", D=o1, **kw))
    t_old = timed(lambda: ops.call("gemm", D=o2, variant=1, **kw))
    ref = a.double() @ w.double().t() + b.double()
    e1 = float((o1[:, :N].double() - ref).norm() / ref.norm())
    e2 = float((o2[:, :N].double() - ref).norm() / ref.norm())
    gb = M * K * 4 / 1e9
    print(f"N={N:2d}: variant {v}: {t_new:7.1f} us ({gb / t_new * 1e3:.2f} TB/s of A), rel err vs fp64 {e1:.1e}  |  register-staged x3: {t_old:7.1f} us, {e2:.1e}", flush=True)
