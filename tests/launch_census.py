"""Launch census of one training step on the CPU emulator: every aten op that would be a device kernel and every C-ABI call, grouped by
the product source line that issued it.  Test infrastructure (uses oracle/abi_emul as the library); run from the repo root:
    python tests/launch_census.py [config=mini_ctr] [prec=bf16]
"""
import collections
import os
import sys
import traceback

import torch
from torch.utils._python_dispatch import TorchDispatchMode

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import conftest  # noqa: E402
import mtt_amd  # noqa: E402
from oracle import abi_emul, configs, weights  # noqa: E402

PKG = os.path.join(ROOT, "multi-task-transformer_amd")
SKIP = {"aten.view.default", "aten._unsafe_view.default", "aten.detach.default", "aten.as_strided.default", "aten.select.int",
        "aten.slice.Tensor", "aten.t.default", "aten.transpose.int", "aten.permute.default", "aten.expand.default", "aten.alias.default",
        "aten.unsqueeze.default", "aten.squeeze.dim", "aten.empty.memory_format", "aten.empty_like.default", "aten.empty_strided.default",
        "aten.reshape.default", "aten.unbind.int", "aten.split.Tensor", "aten.new_empty.default", "aten.lift_fresh.default",
        "aten.is_same_size.default", "aten.unflatten.int", "aten.view_as.default", "aten.squeeze.default", "aten.item.default",
        "aten._local_scalar_dense.default", "aten.result_type.Tensor", "aten.sym_size.int", "aten.stride.int"}


INSIDE = [0]          # > 0 while the emulator runs (its own torch ops are not launches of the product)


def site():
    for fr in reversed(traceback.extract_stack()[:-2]):
        if fr.filename.startswith(PKG):
            return f"{os.path.basename(fr.filename)}:{fr.lineno}"
    return "?"


class Census(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.ops = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if name not in SKIP and not INSIDE[0]:
            self.ops[(name, site())] += 1
        return func(*args, **(kwargs or {}))


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "mini_ctr"
    prec = sys.argv[2] if len(sys.argv) > 2 else "bf16"
    calls = collections.Counter()

    def call(entry, *a, **k):
        calls[(entry, site())] += 1
        INSIDE[0] += 1
        try:
            return abi_emul.call(entry, *a, **k)
        finally:
            INSIDE[0] -= 1
    mtt_amd.ops.call = call
    cfg = configs.taskprompter(name)
    meta, _ = conftest.load_golden(name)
    sd = weights.synth_state_dict(meta["contract"], 0)
    model = conftest.build_product_model(cfg, prec, "cpu")
    model.load_state_dict(sd, strict=True)
    model.train()
    from tests.golden.make_golden import loss_of
    opt = mtt_amd.optim.FusedClipAdam(model.parameters(), lr=1e-4, max_norm=10.0)
    x = weights.synth_images(2, cfg["img_size"], 2)
    for it in range(2):                       # the second step is the steady state (packs rebuilt after the optimizer ran)
        calls.clear()
        with Census() as c:
            opt.zero_grad()
            out = model(x)
            loss_of(out).backward()
            opt.step()
    tot_ops, tot_calls = sum(c.ops.values()), sum(calls.values())
    print(f"# {name} {prec}: {tot_calls} C-ABI calls + {tot_ops} aten ops in step 2  (depth {cfg.get('depth')})")
    by_entry = collections.Counter()
    for (e, s), n in calls.items():
        by_entry[e] += n
    print("# C-ABI calls by entry:", dict(by_entry.most_common()))
    print("# aten ops by (op, site):")
    for (op, s), n in c.ops.most_common(90):
        print(f"{n:6d}  {op:45s} {s}")
    print("# C-ABI calls by (entry, site):")
    for (e, s), n in calls.most_common(60):
        print(f"{n:6d}  {e:45s} {s}")


if __name__ == "__main__":
    main()
