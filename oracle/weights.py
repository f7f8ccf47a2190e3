"""TEST INFRASTRUCTURE ONLY — deterministic synthetic weights / inputs.

Weights are a pure function of (ordered (name, shape) list, seed) through torch's CPU mt19937
generator, which is platform independent: the GPU box regenerates bit-identical tensors from the
state-dict contract alone (tests/golden/*.json), so no multi-MB weight fixtures are committed.
Scales are chosen so that attention logits are O(1..10) (exercises the (1+logit) modulation,
the softmax and the cross-task reweighting far from their trivial operating point) and BN / LN
affine + running stats are non-trivial (folding bugs show).
"""
import math

import torch


def synth_tensor(name, shape, g):
    shape = tuple(shape)
    leaf = name.rsplit(".", 1)[-1]
    if leaf == "num_batches_tracked":
        return torch.zeros(shape, dtype=torch.int64)
    if leaf == "running_mean":
        return torch.randn(shape, generator=g) * 0.1
    if leaf == "running_var":
        return torch.rand(shape, generator=g) * 0.5 + 0.75
    if leaf == "relative_position_bias_table":                 # taskprompter_swin.py:143-144 (learned; O(1) so that it shapes the softmax)
        return torch.randn(shape, generator=g) * 0.5
    if leaf == "task_prompts":
        return torch.randn(shape, generator=g) + 1.0           # taskprompter.py:343-344 (mean 1, std 1)
    if leaf in ("pos_embed", "cls_token"):
        return torch.randn(shape, generator=g) * 0.2
    if leaf == "bias":
        return torch.randn(shape, generator=g) * 0.05
    if leaf == "weight" and len(shape) == 1:                   # LayerNorm / BatchNorm affine
        return torch.rand(shape, generator=g) * 0.5 + 0.75
    if leaf == "weight":
        if "mt_proj.0" in name and len(shape) == 4 and shape[2] == 2:   # ConvTranspose2d [in,out,2,2]
            fan_in = shape[0]
        elif name.endswith("scale_embed.0.weight"):            # ConvTranspose2d [in,out,3,3]
            fan_in = shape[0] * 9 / 4
        else:
            fan_in = 1
            for s in shape[1:]:
                fan_in *= s
        return torch.randn(shape, generator=g) / math.sqrt(fan_in)
    raise KeyError(f"no synthetic rule for {name} {shape}")


DERIVED_BUFFERS = ("relative_position_index", "attn_mask")     # functions of the geometry (taskprompter_swin.py:147-157, 281-300)


def synth_state_dict(contract, seed=0, keep=None):
    """contract: ordered list of (name, shape).  Returns {name: tensor} (fp32, int64 for counters).  Buffers that are pure functions
    of the geometry (DERIVED_BUFFERS) are not synthesised: they are taken from `keep` (a state dict holding the model's own values)
    when given and left out otherwise (load with strict=False)."""
    g = torch.Generator().manual_seed(seed)
    out = {}
    for name, shape in contract:
        if name.rsplit(".", 1)[-1] in DERIVED_BUFFERS:
            if keep is not None:
                out[name] = keep[name].clone()
            continue
        out[name] = synth_tensor(name, shape, g)
    return out


def synth_images(batch, img_size, seed=1):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(batch, 3, img_size[0], img_size[1], generator=g)
