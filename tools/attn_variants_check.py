"""Bitwise comparison of the forward / backward attention kernel variants (mtt_attn_desc.variant) on ragged shapes.
Usage (GPU box): python tools/attn_variants_check.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import mtt_amd  # noqa: E402,F401
from mtt_amd import ops  # noqa: E402

_call = ops.call
FORCE = {"v": 0}
ops.call = lambda name, **kw: _call(name, **(dict(kw, variant=FORCE["v"]) if name == "attn_fwd" else kw))
prec = ops.Prec("bf16")
dev = torch.device("cuda")
bad = 0
for (B, N, nH, T) in [(2, 30, 2, 3), (1, 64, 1, 0), (3, 65, 2, 6), (2, 96, 1, 2), (2, 97, 1, 2), (2, 150, 2, 6), (1, 257, 2, 6), (2, 1030, 16, 6), (1, 8194, 4, 6), (1, 128, 3, 6), (1, 129, 3, 0), (3, 1030, 16, 6), (2, 1056, 16, 6), (2, 1057, 8, 0)]:
    torch.manual_seed(N)
    qkv = (torch.randn(B * N, 3 * nH * 64, device=dev) * 1.5).to(torch.bfloat16)
    FORCE["v"] = 0
    ref = ops.attention(qkv, B, N, nH, T, prec, want_lse=True)
    for v in (2, 3):
        FORCE["v"] = v
        got = ops.attention(qkv, B, N, nH, T, prec, want_lse=True)
        for _ in range(4):                                  # repeats: a staging race would show as run-to-run differences
            again = ops.attention(qkv, B, N, nH, T, prec, want_lse=True)
            if not torch.equal(again[0], got[0]):
                bad += 1
                print(f"NOT REPEATABLE B={B} N={N} nH={nH} T={T} variant {v}")
                break
        ok = all((a is None and b is None) or torch.equal(a, b) for a, b in zip(ref, got))
        fin = bool(torch.isfinite(got[0].float()).all())
        if not (ok and fin):
            bad += 1
            d = (ref[0].float() - got[0].float()).abs().max().item()
            print(f"MISMATCH B={B} N={N} nH={nH} T={T} variant {v}: max |out diff| {d:.3e}, finite {fin}")
    # backward: dq / dkv kernels of every variant against the default, same forward outputs
    FORCE["v"] = 0
    ao, rawlog, lse = ref
    dao = torch.randn(B * N, nH * 64, device=dev).to(torch.bfloat16)
    drawlog = torch.randn(B, nH, T, N, device=dev) * 0.01 if T else None
    dsum = torch.empty(B, nH, 2, (N + 3) // 4 * 4, device=dev)
    outs = {}
    for v in (0, 3, 2):
        dq = torch.full_like(qkv, float("nan"))
        for rep in range(3):
            _call("attn_bwd", qkv=qkv, out=ao, rawlog=None, lse=lse, B=B, N=N, nH=nH, T=T, dtype=1, prec=0, scale=0.125, variant=v,
                  xargs=[dao, drawlog, dq, dsum])
            if rep and not torch.equal(dq, outs[v]):
                bad += 1
                print(f"BACKWARD NOT REPEATABLE B={B} N={N} nH={nH} T={T} variant {v}")
            outs[v] = dq.clone()
        if not torch.equal(outs[v], outs[0]) or not bool(torch.isfinite(outs[v].float()).all()):
            bad += 1
            print(f"BACKWARD MISMATCH B={B} N={N} nH={nH} T={T} variant {v}: max |diff| {(outs[v].float() - outs[0].float()).abs().max().item():.3e}")
    print(f"B={B} N={N} nH={nH} T={T}: checked")
print("variants check:", "OK" if bad == 0 else f"{bad} mismatches")
sys.exit(1 if bad else 0)
