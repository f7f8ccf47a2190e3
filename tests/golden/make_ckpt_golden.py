"""Golden fixture for the checkpoint importers (multi-task-transformer_amd/checkpoints.py): a synthetic Flax-layout ViT `.npz`
(4 blocks, C = 64, source grid 3x3 + class slot) is loaded by the UNMODIFIED reference loaders
(TaskPrompter/models/transformers/taskprompter.py:385 load_pretrained / InvPT/models/transformers/vit.py _load_weights) into the
reference's miniature models, whose grids differ (position embeddings get resized); the resulting backbone tensors are the
expected values.  Also: the reference's checkpoint_filter_fn on a torch state dict with a flattened patch embedding and a
3x3-grid pos_embed.   Run here (needs /root/reference):  python tests/golden/make_ckpt_golden.py
"""
import os
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import configs, ref_build, ref_import  # noqa: E402


STRIDE = 5


def fake_flax_vit(C, depth, heads, grid, seed=0):
    rng = np.random.RandomState(seed)
    hd = C // heads
    w = {"embedding/kernel": rng.randn(16, 16, 3, C), "embedding/bias": rng.randn(C), "cls": rng.randn(1, 1, C),
         "Transformer/posembed_input/pos_embedding": rng.randn(1, 1 + grid * grid, C),
         "Transformer/encoder_norm/scale": rng.randn(C), "Transformer/encoder_norm/bias": rng.randn(C)}
    for i in range(depth):
        b = f"Transformer/encoderblock_{i}/"
        m = b + "MultiHeadDotProductAttention_1/"
        for n in ("query", "key", "value"):
            w[m + n + "/kernel"] = rng.randn(C, heads, hd)
            w[m + n + "/bias"] = rng.randn(heads, hd)
        w[m + "out/kernel"] = rng.randn(heads, hd, C)
        w[m + "out/bias"] = rng.randn(C)
        for k in ("LayerNorm_0", "LayerNorm_2"):
            w[b + k + "/scale"] = rng.randn(C)
            w[b + k + "/bias"] = rng.randn(C)
        w[b + "MlpBlock_3/Dense_0/kernel"] = rng.randn(C, 4 * C)
        w[b + "MlpBlock_3/Dense_0/bias"] = rng.randn(4 * C)
        w[b + "MlpBlock_3/Dense_1/kernel"] = rng.randn(4 * C, C)
        w[b + "MlpBlock_3/Dense_1/bias"] = rng.randn(C)
    return {k: v.astype(np.float32) for k, v in w.items()}


def main():
    out = {}
    for kind, name in (("TP", "mini_ctr"), ("IP", "mini")):
        cfg = dict(configs.taskprompter(name) if kind == "TP" else configs.invpt(name), backbone="nano")
        C, depth, heads, _ = configs.VIT[cfg["backbone"]]
        flax = fake_flax_vit(C, depth, heads, 3, seed=1 if kind == "TP" else 2)
        model, _ = ref_build.build_reference(cfg, randomize=False)
        backbone = model.backbone
        with tempfile.TemporaryDirectory() as td:
            path = os.path.join(td, "vit.npz")
            np.savez(path, **flax)
            if kind == "TP":
                backbone.load_pretrained(path)
            else:
                type(backbone).__init__.__globals__["_load_weights"](backbone, path)
        # the synthetic checkpoint itself is NOT stored: tests regenerate it with fake_flax_vit (numpy RandomState is portable)
        covered = [k for k, _ in backbone.named_parameters()
                   if k.split(".")[0] in ("patch_embed", "pos_embed", "cls_token", "norm", "blocks")
                   and not any(s in k for s in ("token_trans", "chan_"))]
        sd = backbone.state_dict()
        for k in covered:                    # every STRIDE-th element (random data: any layout mistake shows), pos_embed in full
            out[f"{kind}/expect/{k}"] = sd[k].numpy() if k == "pos_embed" else sd[k].numpy().reshape(-1)[::STRIDE]
        if kind == "TP":                     # checkpoint_filter_fn on a torch state dict
            filter_fn = type(backbone).__init__.__globals__["checkpoint_filter_fn"]
            g = torch.Generator().manual_seed(3)
            raw = {"model": {"patch_embed.proj.weight": torch.randn(C, 3 * 16 * 16, generator=g),
                             "pos_embed": torch.randn(1, 10, C, generator=g), "norm.bias": torch.randn(C, generator=g)}}
            filt = filter_fn(raw, backbone)
            for k, v in raw["model"].items():
                out[f"TP/filter_in/{k}"] = v.numpy()
            for k, v in filt.items():
                out[f"TP/filter_out/{k}"] = v.numpy()
    np.savez_compressed(os.path.join(HERE, "ckpt_import.npz"), **out)
    print("wrote", os.path.join(HERE, "ckpt_import.npz"), sum(v.size for v in out.values()) * 4 / 1e6, "MB")


if __name__ == "__main__":
    main()
