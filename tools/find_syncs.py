"""Runs two training steps with torch.cuda.set_sync_debug_mode('warn'): every host<->device synchronisation on the step's path
prints a warning with its Python stack (a sync costs the CPU its run-ahead over the GPU)."""
import os
import sys
import warnings

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import mtt_amd  # noqa: E402

B = 4
dev = torch.device("cuda")
p = mtt_amd.factory.make_p(mtt_amd.factory.TASK_ORDER, (512, 512), backbone="TaskPrompter_vitL", head="conv", embed_dim=300,
                           final_embed_dim=350, chan_nheads=1, use_ctr=True, prec="bf16")
model = mtt_amd.factory.get_model(p).to(dev).train()
crit = mtt_amd.losses.FusedMultiTaskLoss(p, p.TASKS.NAMES).to(dev)
opt = mtt_amd.optim.FusedClipAdam(model.parameters(), lr=2e-5, weight_decay=1e-6, max_norm=10.0)
x = torch.randn(B, 3, 512, 512, device=dev)
gt = mtt_amd.losses.synthetic_targets(p, B, 512, 512, dev)


def step():
    loss = crit(model(x), gt)["total"]
    opt.zero_grad(set_to_none=True)
    loss.backward()
    opt.step()


step()
torch.cuda.synchronize()
warnings.simplefilter("always")
torch.cuda.set_sync_debug_mode("warn")
step()
step()
torch.cuda.set_sync_debug_mode("default")
torch.cuda.synchronize()
print("done")
