"""Weight-gradient paths on the NS-6 shapes at per-GPU batch 63: round-1 (transposing copies of both operands + K-contiguous LDS-DMA
GEMM) vs the token-major kernel (gemm_tn_kernel, no copies), and the 3x3 conv weight gradient (register-staged transposing kernel vs
gemm_tn_kernel with the implicit im2col^T).  Times include everything each path launches (copies, slab sums, bias column sums)."""
import importlib
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import mtt_amd  # noqa: E402
from mtt_amd import ops  # noqa: E402

ap = importlib.import_module("multi-task-transformer_amd.autograd_path")
prec = ops.Prec("bf16")
M63 = 63 * 1030


def timed(fn, rounds=4, inner=2):
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


for name, N, Kp in (("qkv", 3072, 1024), ("proj", 1024, 1024), ("fc1", 4096, 1024), ("fc2", 1024, 4096)):
    dy = (torch.rand(M63, N, device="cuda") * 2 - 1).bfloat16()
    x = (torch.rand(M63, Kp, device="cuda") * 2 - 1).bfloat16()
    res = {}
    for tn in (False, True):
        ap.WGRAD_TN = tn
        res[tn] = timed(lambda: ap._enc_wgrad(dy, x, N, Kp, prec))
    ap.WGRAD_TN = False
    a, _ = ap._enc_wgrad(dy, x, N, Kp, prec)
    ap.WGRAD_TN = True
    b, _ = ap._enc_wgrad(dy, x, N, Kp, prec)
    err = float((a - b).norm() / a.norm())
    fl = 2.0 * M63 * N * Kp
    print(f"wgrad {name:5s} N={N} Kp={Kp}: transposes + DMA GEMM {res[False] * 1e3:7.1f} us ({fl / res[False] / 1e9:5.0f} TF/s)  |  "
          f"token-major kernel {res[True] * 1e3:7.1f} us ({fl / res[True] / 1e9:5.0f} TF/s)   rel diff {err:.1e}", flush=True)

_call = ops.call
FORCE = {"v": 0}
ops.call = lambda name, **kw: _call(name, **(dict(kw, variant=FORCE["v"]) if name == "gemm" else kw))
Z, C = 6, 350
Cp = ops.pad8(C)
for H, B in ((128, 16), (32, 63)):
    rows = B * H * H
    x = torch.randn(Z, rows, Cp, device="cuda").bfloat16(); x[..., C:] = 0
    dy = torch.randn(Z, rows, Cp, device="cuda").bfloat16(); dy[..., C:] = 0
    conv = dict(H=H, W=H, C=C, Cp=Cp, dil=1, flip=0)
    S = 4 if B % 4 == 0 else 3
    c = rows // S
    slabs = torch.empty(Z, S, C, 9 * Cp, device="cuda")

    def run():
        ap._gemm(dy, x, slabs, C, 9 * Cp, c, prec, a_op=ap.OP_R, b_op=ap.OP_CONV_R, lda=Cp, ldb=Cp, ldd=9 * Cp, batch=Z * S, batch_inner=S,
                 a_zo=rows * Cp, a_zi=c * Cp, b_zo=rows * Cp, b_zi=c * Cp, d_zo=S * C * 9 * Cp, d_zi=C * 9 * Cp, conv=conv)
        return slabs.sum(1)
    out = {}
    for v, nm in ((1, "register-staged"), (0, "policy (token-major kernel)")):
        FORCE["v"] = v
        t = timed(run, rounds=3, inner=1)
        out[v] = (t, run())
    fl = 2.0 * rows * C * 9 * C * Z
    err = float((out[0][1] - out[1][1]).norm() / out[1][1].norm())
    print(f"conv3x3 wgrad B={B} {H}x{H} C={C} Z={Z} ({S} image slices): register-staged {out[1][0]:.3f} ms ({fl / out[1][0] / 1e9:4.0f} TF/s)  |  "
          f"token-major kernel {out[0][0]:.3f} ms ({fl / out[0][0] / 1e9:4.0f} TF/s)   rel diff {err:.1e}", flush=True)
