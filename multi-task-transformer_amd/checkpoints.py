"""On-disk formats either side of the hot path (SURVEY.md §8 f rank 2): importing a Google/Flax ViT `.npz` checkpoint into the
native modules (what the reference's `load_pretrained` -> `_load_weights` does, TaskPrompter/models/transformers/taskprompter.py:385,
:525-602 and InvPT/models/transformers/vit.py:410-488), resizing position embeddings to another input size (:605-624) and filtering
a torch state dict (:627-643).  Host-side, one-off work: plain torch / numpy on the CPU, nothing here is on the timed path.

Table driven: `flax_vit_plan(model)` lists (parameter name, npz key(s), layout rule) triples; `load_flax_vit_npz` applies them.
Layout rules (Flax kernels are stored input-major):
    vec      copy as is
    conv     HWIO -> OIHW                      (patch embedding)
    dense    [in, out] -> [out, in]
    qkv      three [C, heads, hd] kernels -> one [3C, C] weight (rows: all q, then k, then v)
    qkv_bias three [heads, hd] biases -> [3C]
    proj     [heads, hd, C] -> [C, heads*hd]
    pos      [1, 1 + g*g, C] resized bicubically to the model's grid when the sizes differ
"""
import math

import numpy as np
import torch
import torch.nn.functional as F


def resize_pos_embed(posemb, n_new, num_tokens=1, grid_new=None):
    """[1, num_tokens + g*g, C] -> [1, n_new, C]: the grid part is resampled bicubically (align_corners=False) to `grid_new`
    (h, w) (default: square), the leading `num_tokens` class/distillation slots are kept."""
    posemb = torch.as_tensor(posemb)
    tok, grid = posemb[:, :num_tokens], posemb[0, num_tokens:]
    g_old = int(math.isqrt(grid.shape[0]))
    assert g_old * g_old == grid.shape[0], "source position embedding grid must be square"
    if grid_new is None or not len(grid_new):
        side = int(math.isqrt(n_new - num_tokens))
        grid_new = (side, side)
    gh, gw = int(grid_new[0]), int(grid_new[1])
    assert gh * gw == n_new - num_tokens
    img = grid.reshape(1, g_old, g_old, -1).permute(0, 3, 1, 2)
    img = F.interpolate(img, size=(gh, gw), mode="bicubic", align_corners=False)
    return torch.cat([tok, img.permute(0, 2, 3, 1).reshape(1, gh * gw, -1)], dim=1)


def _grid_of(model):
    pe = model.patch_embed
    if hasattr(pe, "grid_size"):
        return tuple(pe.grid_size)
    ih, iw = pe.img_size
    return (ih // 16, iw // 16)


def flax_vit_plan(model, prefix=""):
    """[(parameter name, npz key or tuple of keys, rule)] for a ViT encoder (TaskPrompter backbone or InvPT's VisionTransformer)."""
    t = prefix + "Transformer/"
    plan = [("patch_embed.proj.weight", prefix + "embedding/kernel", "conv"),
            ("patch_embed.proj.bias", prefix + "embedding/bias", "vec"),
            ("pos_embed", t + "posembed_input/pos_embedding", "pos"),
            ("norm.weight", t + "encoder_norm/scale", "vec"), ("norm.bias", t + "encoder_norm/bias", "vec")]
    if hasattr(model, "cls_token"):                      # InvPT's ViT keeps the class token; TaskPrompter drops it (only its pos slot stays)
        plan.append(("cls_token", prefix + "cls", "vec"))
    for i in range(len(model.blocks)):
        b, m = f"{t}encoderblock_{i}/", f"{t}encoderblock_{i}/MultiHeadDotProductAttention_1/"
        plan += [(f"blocks.{i}.norm1.weight", b + "LayerNorm_0/scale", "vec"), (f"blocks.{i}.norm1.bias", b + "LayerNorm_0/bias", "vec"),
                 (f"blocks.{i}.attn.qkv.weight", tuple(m + n + "/kernel" for n in ("query", "key", "value")), "qkv"),
                 (f"blocks.{i}.attn.qkv.bias", tuple(m + n + "/bias" for n in ("query", "key", "value")), "qkv_bias"),
                 (f"blocks.{i}.attn.proj.weight", m + "out/kernel", "proj"), (f"blocks.{i}.attn.proj.bias", m + "out/bias", "vec"),
                 (f"blocks.{i}.norm2.weight", b + "LayerNorm_2/scale", "vec"), (f"blocks.{i}.norm2.bias", b + "LayerNorm_2/bias", "vec")]
        for r in range(2):
            plan += [(f"blocks.{i}.mlp.fc{r + 1}.weight", b + f"MlpBlock_3/Dense_{r}/kernel", "dense"),
                     (f"blocks.{i}.mlp.fc{r + 1}.bias", b + f"MlpBlock_3/Dense_{r}/bias", "vec")]
    return plan


def _apply_rule(rule, arrs, target, model):
    a = [torch.from_numpy(np.asarray(x)) for x in arrs]
    if rule == "vec":
        return a[0].reshape(target.shape)
    if rule == "conv":
        w = a[0].permute(3, 2, 0, 1)                               # HWIO -> OIHW
        if w.shape[1] != target.shape[1]:                          # grey-scale / other channel counts: like timm's adapt_input_conv
            if target.shape[1] == 1:
                w = w.sum(1, keepdim=True)
            else:
                rep = -(-target.shape[1] // 3)
                w = w.repeat(1, rep, 1, 1)[:, :target.shape[1]] * (3.0 / target.shape[1])
        return w
    if rule == "dense":
        return a[0].t()
    if rule == "qkv":
        return torch.cat([x.reshape(x.shape[0], -1).t() for x in a], 0)
    if rule == "qkv_bias":
        return torch.cat([x.reshape(-1) for x in a], 0)
    if rule == "proj":
        return a[0].reshape(-1, a[0].shape[-1]).t()                # [heads, hd, C] -> [C, heads*hd]
    if rule == "pos":
        w = a[0]
        if tuple(w.shape) != tuple(target.shape):
            w = resize_pos_embed(w, target.shape[1], getattr(model, "num_tokens", 1), _grid_of(model))
        return w
    raise ValueError(rule)


@torch.no_grad()
def load_flax_vit_npz(model, checkpoint, prefix=""):
    """Copy a Flax ViT checkpoint (path to a .npz or an already opened mapping) into `model` (a TaskPrompter backbone or InvPT's
    VisionTransformer).  Parameters the checkpoint does not cover (task prompts, channel-attention, decoders, heads) are left as
    initialised.  Returns the list of parameter names that were loaded."""
    w = np.load(checkpoint) if isinstance(checkpoint, (str, bytes)) or hasattr(checkpoint, "__fspath__") else checkpoint
    if not prefix and "opt/target/embedding/kernel" in w:
        prefix = "opt/target/"
    if hasattr(model.patch_embed, "backbone"):
        raise NotImplementedError("hybrid (ResNet stem) ViT checkpoints are not used by the reference's configs")
    params = dict(model.named_parameters())
    done = []
    for name, keys, rule in flax_vit_plan(model, prefix):
        keys = keys if isinstance(keys, tuple) else (keys,)
        target = params[name]
        val = _apply_rule(rule, [w[k] for k in keys], target, model)
        if tuple(val.shape) != tuple(target.shape):
            raise ValueError(f"{name}: checkpoint gives {tuple(val.shape)}, model has {tuple(target.shape)}")
        target.copy_(val.to(target.dtype))
        done.append(name)
    return done


# `pretrained=True` (what the reference's get_backbone passes, TaskPrompter/utils/common_config.py:22,29; InvPT/utils/common_config.py:17):
# timm 0.5.4's build_model_with_cfg -> load_custom_pretrained fetches default_cfg['url'] into the torch hub cache
# (<torch.hub.get_dir()>/checkpoints/<file name of the URL>) and hands the file to model.load_pretrained.  This box has no network, so the
# native path takes the SAME cache location and fails with the path to fill when the file is not there.
_AUGREG = "https://storage.googleapis.com/vit_models/augreg/"
PRETRAINED_URLS = {            # default_cfgs of taskprompter.py:83-91 / vit.py (the variants the reference's factories construct)
    "vit_base_patch16_384": _AUGREG + "B_16-i21k-300ep-lr_0.001-aug_medium1-wd_0.1-do_0.0-sd_0.0--imagenet2012-steps_20k-lr_0.01-res_384.npz",
    "vit_large_patch16_384": _AUGREG + "L_16-i21k-300ep-lr_0.001-aug_medium1-wd_0.1-do_0.1-sd_0.1--imagenet2012-steps_20k-lr_0.01-res_384.npz",
    "vit_small_patch16_384": _AUGREG + "S_16-i21k-300ep-lr_0.001-aug_light1-wd_0.03-do_0.0-sd_0.0--imagenet2012-steps_20k-lr_0.03-res_384.npz",
}


def cached_pretrained_path(variant):
    """Where timm's download_cached_file keeps the variant's checkpoint: <hub dir>/checkpoints/<basename of the URL>."""
    import os
    from urllib.parse import urlparse
    url = PRETRAINED_URLS[variant]
    return os.path.join(torch.hub.get_dir(), "checkpoints", os.path.basename(urlparse(url).path))


def load_cached_pretrained(model, variant):
    """pretrained=True: load the variant's Flax `.npz` from the torch hub cache (never downloads)."""
    import os
    if variant not in PRETRAINED_URLS:
        raise RuntimeError(f"no pretrained checkpoint is defined for variant {variant!r}")
    path = cached_pretrained_path(variant)
    if not os.path.isfile(path):
        raise RuntimeError(f"pretrained=True needs {path} (timm's cache of {PRETRAINED_URLS[variant]}); this library never downloads — "
                           "place the file there, or construct with pretrained=False and load a checkpoint with load_state_dict")
    return model.load_pretrained(path)


def filter_state_dict(state_dict, model):
    """A torch checkpoint made for another input size / an old linear patch embedding, adapted to `model`: unwraps {'model': ...},
    reshapes a flattened patch-embedding weight to OIHW and resizes `pos_embed`."""
    sd = state_dict["model"] if "model" in state_dict else state_dict
    out = {}
    for k, v in sd.items():
        if "patch_embed.proj.weight" in k and v.dim() < 4:
            o, _, kh, kw = model.patch_embed.proj.weight.shape
            v = v.reshape(o, -1, kh, kw)
        elif k == "pos_embed" and tuple(v.shape) != tuple(model.pos_embed.shape):
            v = resize_pos_embed(v, model.pos_embed.shape[1], getattr(model, "num_tokens", 1), _grid_of(model))
        out[k] = v
    return out
