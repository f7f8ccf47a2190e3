"""Golden fixture of the wrapper's `dd_label_map_size` branch (TaskPrompter/models/taskprompter_wrapper.py:17-27: predictions resized to
a configured label-map size instead of the input size), generated from the UNMODIFIED reference in the build container:

    python tests/golden/make_dd_golden.py

The reference TaskPrompterWrapper of the `mini_ctr` miniature is built with `p.dd_label_map_size = (40, 56)` (neither the 64 x 96 input
nor a multiple of the 16 x 24 head maps), loaded with the deterministic synthetic weights (oracle/weights.py, seed 0) and run in eval
mode on the seeded batch (seed 1) -> tests/golden/mini_ctr_dd.npz {eval/<task>}."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import configs, ref_build, ref_import, weights  # noqa: E402

DD_SIZE = (40, 56)


def main():
    cfg = configs.taskprompter("mini_ctr")
    model, p = ref_build.build_reference(cfg, randomize=False)
    p.dd_label_map_size = DD_SIZE
    ns = ref_import.load_reference("TP")
    model = ns.TaskPrompterWrapper(p, model.backbone, model.heads)           # the constructor reads the key (taskprompter_wrapper.py:17-20)
    assert model.target_size == DD_SIZE
    contract = [(k, list(v.shape)) for k, v in model.state_dict().items()]
    model.load_state_dict(weights.synth_state_dict(contract, seed=0, keep=model.state_dict()), strict=True)
    model.eval()
    with torch.no_grad():
        out = model(weights.synth_images(2, cfg["img_size"], seed=1))
    arrays = {f"eval/{k}": v.numpy().astype(np.float32) for k, v in out.items()}
    assert all(tuple(v.shape[-2:]) == DD_SIZE for v in arrays.values())
    np.savez_compressed(os.path.join(HERE, "mini_ctr_dd.npz"), **arrays)
    print("mini_ctr_dd", {k: v.shape for k, v in arrays.items()}, os.path.getsize(os.path.join(HERE, "mini_ctr_dd.npz")), "bytes")


if __name__ == "__main__":
    main()
