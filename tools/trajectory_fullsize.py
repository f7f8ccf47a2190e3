"""Loss trajectories of the default mixed-precision mode (x3f: fp32-class forward, bf16 backward) and of the fully fp32-class mode (x3:
gradients pinned to the CPU oracle's autograd at 6e-5) at the BENCHMARK's model size: TaskPrompter ViT-L, 512 x 512, 6 tasks, from the same
initial weights on the same cycle of synthetic batches — forward + FusedMultiTaskLoss + backward + clip 10 + Adam, DropPath off so that
both runs see the same computation.  The miniature-model test with the CPU oracle in the loop is
tests/test_gpu_train.py::test_mixed_precision_training_trajectory_...; this tool is its full-size companion (no oracle: 9 s per CPU step).
Usage: python tools/trajectory_fullsize.py [steps=200] [batch=8] [lr=2e-4]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import mtt_amd  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
lr = float(sys.argv[3]) if len(sys.argv) > 3 else 2e-4
TASKS = ["semseg", "depth", "human_parts", "sal", "normals", "edge"]
H = W = 512
dev = torch.device("cuda", 0)
NB = 4
xs = [torch.randn(B, 3, H, W, generator=torch.Generator().manual_seed(100 + i)).to(dev) for i in range(NB)]
curves, sd0 = {}, None
for mode in ("x3f", "x3"):
    torch.manual_seed(0)
    p = mtt_amd.factory.make_p(TASKS, (H, W), backbone="TaskPrompter_vitL", head="conv", embed_dim=300, final_embed_dim=350, chan_nheads=1, use_ctr=True,
                               prec=mode, drop_path_rate=0.0)
    model = mtt_amd.factory.get_model(p).to(dev)
    if sd0 is None:
        sd0 = {k: v.detach().clone() for k, v in model.state_dict().items()}
    else:
        model.load_state_dict(sd0)
    model.train()
    crit = mtt_amd.losses.FusedMultiTaskLoss(p, p.TASKS.NAMES).to(dev)
    gts = [mtt_amd.losses.synthetic_targets(p, B, H, W, dev, seed=200 + i) for i in range(NB)]
    opt = mtt_amd.optim.FusedClipAdam(model.parameters(), lr=lr, weight_decay=1e-6, max_norm=10.0)
    ls = []
    for it in range(steps):
        loss = crit(model(xs[it % NB]), gts[it % NB])["total"]
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        ls.append(loss.detach())
    curves[mode] = [float(v) for v in torch.stack(ls).cpu()]
    del model, opt, crit, gts
    mtt_amd.ops.clear_pack_cache()
    torch.cuda.empty_cache()
a, b = torch.tensor(curves["x3f"]), torch.tensor(curves["x3"])
k = 10
ra, rb = a.unfold(0, k, 1).mean(1), b.unfold(0, k, 1).mean(1)
print(json.dumps(dict(what="NS-6 (ViT-L, 512x512, 6 tasks) training trajectories, x3f vs x3, same initial state and batches, DropPath off", steps=steps, batch=B, lr=lr,
                      first=curves["x3"][0], x3_last10=float(rb[-1]), x3f_last10=float(ra[-1]),
                      max_pointwise_gap=float(((a - b).abs() / b.abs()).max()), max_running10_gap=float(((ra - rb).abs() / rb.abs()).max()),
                      final_running10_gap=float((ra[-1] - rb[-1]).abs() / rb[-1].abs()),
                      every20={m: [round(c[i], 4) for i in range(0, steps, 20)] for m, c in curves.items()})))
