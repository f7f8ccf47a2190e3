"""Per-step wall time of the first training iterations (does the step reach its steady state within bench.py's default 2 warm-up steps?).
Usage (GPU box): python tools/step_times.py [config] [prec] [n_steps]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
import mtt_amd  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "ns6"
prec = sys.argv[2] if len(sys.argv) > 2 else "x3f"
n = int(sys.argv[3]) if len(sys.argv) > 3 else 14
_, _, (H, W), batch, _ = bench.CONFIGS[cfg]
p, model = bench.build(cfg, prec, mtt_amd)
dev = torch.device("cuda")
model = model.to(dev).train()
crit = mtt_amd.losses.FusedMultiTaskLoss(p, p.TASKS.NAMES).to(dev)
opt = mtt_amd.optim.FusedClipAdam(model.parameters(), lr=2e-5, weight_decay=1e-6, max_norm=10.0)
x = torch.randn(batch, 3, H, W, device=dev)
gt = mtt_amd.losses.synthetic_targets(p, batch, H, W, dev)
times = []
for i in range(n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    loss = crit(model(x), gt)["total"]
    opt.zero_grad(set_to_none=True)
    loss.backward()
    opt.step()
    torch.cuda.synchronize()
    times.append((time.perf_counter() - t0) * 1e3)
    print(f"step {i:2d}: {times[-1]:8.1f} ms   allocated {torch.cuda.memory_allocated() / 2**30:6.1f} GiB  reserved {torch.cuda.memory_reserved() / 2**30:6.1f} GiB", flush=True)
