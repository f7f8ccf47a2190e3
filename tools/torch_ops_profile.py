"""Which torch (non-libmtt) ops run inside one training step, with input shapes: finds glue that should be fused or removed.
Usage (GPU box): python tools/torch_ops_profile.py [batch] [prec] [bench config]   (the product's own criterion and optimizer, as bench.py's step)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

import mtt_amd  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
PREC = sys.argv[2] if len(sys.argv) > 2 else "x3f"
CONFIG = sys.argv[3] if len(sys.argv) > 3 else "ns6"          # a bench.py config name (ns6, cfg2 ... swinb)
dev = torch.device("cuda")
import bench  # noqa: E402
_, kw, (H, W), _, _ = bench.CONFIGS[CONFIG]
kw = dict(kw)
p = mtt_amd.factory.make_p(kw.pop("tasks"), (H, W), prec=PREC, **kw)
model = mtt_amd.factory.get_model(p).to(dev).train()
crit = mtt_amd.losses.FusedMultiTaskLoss(p, p.TASKS.NAMES).to(dev)
opt = mtt_amd.optim.FusedClipAdam(model.parameters(), lr=2e-5, weight_decay=1e-6, max_norm=10.0)
x = torch.randn(B, 3, H, W, device=dev)
gt = mtt_amd.losses.synthetic_targets(p, B, H, W, dev)


def step():
    loss = crit(model(x), gt)["total"]
    opt.zero_grad(set_to_none=True)
    loss.backward()
    opt.step()


step()
step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    step()
    torch.cuda.synchronize()
rows = [e for e in prof.key_averages(group_by_input_shape=True) if e.key.startswith("aten::") and e.device_time_total > 0]
rows.sort(key=lambda e: -e.device_time_total)
print("aten ops with device time in one training step (B = %d, %s), by input shape:" % (B, PREC))
tot = 0.0
for e in rows[:90]:
    tot += e.device_time_total
    print("%9.1f us  x%-4d  %-28s %s" % (e.device_time_total, e.count, e.key, str(e.input_shapes)[:150]))
print("sum of the listed rows: %.2f ms" % (tot / 1e3))
