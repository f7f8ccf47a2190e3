"""One representative launch set of the dominant kernel for PMC collection (fc1 shape of the NS-6 step at per-GPU batch 40)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import mtt_amd  # noqa: E402
from mtt_amd import ops  # noqa: E402

prec = ops.Prec("bf16")
M, N, K = 41200, 4096, 1024
x = torch.randn(M, K, device="cuda").bfloat16()
w = torch.randn(1, N, K, device="cuda").bfloat16()
b = torch.randn(1, N, device="cuda")
out = torch.empty(1, M, N, device="cuda", dtype=torch.bfloat16)
for _ in range(3):
    ops.linear(x, w, N, prec, bias=b, act=1, out=out)
torch.cuda.synchronize()
print("algorithmic bytes per launch:", (M * K + N * K + M * N) * 2)
