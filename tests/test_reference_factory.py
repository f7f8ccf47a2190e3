"""The drop-in boundary, one level up: the reference's OWN, UNMODIFIED factory
`utils/common_config.get_model(p)` (TaskPrompter/utils/common_config.py:76-90, InvPT/utils/common_config.py:39-51) builds the model
twice — once from the reference's model files, once with the dotted module paths it imports
(`models.transformers.taskprompter`, `models.taskprompter_wrapper` / `models.transformers.vit`, `models.transformers.transformer_decoder`,
`models.transformer_net`) overlaid by this package in `sys.modules`, which is what the import-line edits of INTEGRATION.md section 1
amount to.  The factory passes `pretrained=True`, so both constructions read the SAME synthetic Flax `.npz` from the torch hub cache
(timm's download_cached_file location).  Then: identical state-dict contract, identical imported encoder weights, the reference's
state dict loads strict, and the forward (C-ABI calls on the CPU emulator) matches the reference module's forward per task head.

CPU only, and only where /root/reference exists (the build container): nothing under -m gpu reads the reference."""
import importlib
import os
import sys
import types

import numpy as np
import pytest
import torch

import conftest  # noqa: F401
from oracle import ref_build, ref_import

pytestmark = pytest.mark.skipif(not ref_import.reference_available(), reason="needs the reference checkout (build container only)")

SHIM = os.path.join(os.path.dirname(os.path.abspath(ref_import.__file__)), "_refshim")


def _import_factory(sub, overlay):
    """the unmodified <sub>/utils/common_config.py with `overlay` = {dotted module name: module} pre-seeded in sys.modules"""
    _cleanup()
    sys.path[:0] = [SHIM, os.path.join(ref_import.REFERENCE_ROOT, sub)]
    for name, mod in overlay.items():
        parts = name.split(".")
        for i in range(1, len(parts)):                       # parent packages of the overlaid leaves: empty stand-ins
            sys.modules.setdefault(".".join(parts[:i]), types.ModuleType(".".join(parts[:i])))
        sys.modules[name] = mod
    return importlib.import_module("utils.common_config")


def _cleanup():
    for q in (SHIM, os.path.join(ref_import.REFERENCE_ROOT, "TaskPrompter"), os.path.join(ref_import.REFERENCE_ROOT, "InvPT")):
        while q in sys.path:
            sys.path.remove(q)
    ref_import._purge()


def _write_flax_vit(path, C, depth, heads, grid, seed):
    """A Flax-layout ViT checkpoint (the key set of the augreg `.npz` files the reference downloads, vit.py:410-488); every tensor is
    tiled from 4099 random numbers so that the file compresses while any layout / transposition mistake still shows."""
    rng = np.random.RandomState(seed)
    hd = C // heads

    def t(shape, std, mean=0.0):
        base = (mean + rng.randn(4099) * std).astype(np.float32)
        return np.resize(base, int(np.prod(shape))).reshape(shape)
    w = {"embedding/kernel": t((16, 16, 3, C), 0.02), "embedding/bias": t((C,), 0.1), "cls": t((1, 1, C), 0.1),
         "Transformer/posembed_input/pos_embedding": t((1, 1 + grid * grid, C), 0.1),
         "Transformer/encoder_norm/scale": t((C,), 0.1, 1.0), "Transformer/encoder_norm/bias": t((C,), 0.1)}
    for i in range(depth):
        b = f"Transformer/encoderblock_{i}/"
        m = b + "MultiHeadDotProductAttention_1/"
        for n in ("query", "key", "value"):
            w[m + n + "/kernel"], w[m + n + "/bias"] = t((C, heads, hd), 0.03), t((heads, hd), 0.1)
        w[m + "out/kernel"], w[m + "out/bias"] = t((heads, hd, C), 0.03), t((C,), 0.1)
        for k in ("LayerNorm_0", "LayerNorm_2"):
            w[b + k + "/scale"], w[b + k + "/bias"] = t((C,), 0.1, 1.0), t((C,), 0.1)
        w[b + "MlpBlock_3/Dense_0/kernel"], w[b + "MlpBlock_3/Dense_0/bias"] = t((C, 4 * C), 0.03), t((4 * C,), 0.1)
        w[b + "MlpBlock_3/Dense_1/kernel"], w[b + "MlpBlock_3/Dense_1/bias"] = t((4 * C, C), 0.02), t((C,), 0.1)
    os.makedirs(os.path.dirname(path), exist_ok=True)
    np.savez_compressed(path, **w)


def _rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())


def _compare(ref_model, our_model, size, monkeypatch):
    import mtt_amd
    from oracle import abi_emul
    rs, osd = ref_model.state_dict(), our_model.state_dict()
    assert [(k, tuple(v.shape)) for k, v in rs.items()] == [(k, tuple(v.shape)) for k, v in osd.items()]
    # pretrained=True went through BOTH loaders: the imported encoder tensors are equal before any state dict is copied
    imported = [k for k in rs if k.startswith("backbone.") and "token_trans" not in k
                and any(s in k for s in (".qkv.", ".attn.proj.", ".fc1.", ".fc2.", "patch_embed.", "pos_embed", ".norm1.", ".norm2.", "backbone.norm."))]
    assert len(imported) > 100
    for k in imported:
        assert _rel(osd[k], rs[k]) < 1e-6, k
    ref_build.randomize_norm_state(ref_model)
    our_model.load_state_dict(ref_model.state_dict(), strict=True)
    monkeypatch.setattr(mtt_amd.ops, "call", abi_emul.call)
    mtt_amd.ops.clear_pack_cache()
    x = torch.randn(2, 3, *size, generator=torch.Generator().manual_seed(5))
    ref_model.eval()
    our_model.eval()
    with torch.no_grad():
        want = ref_model(x)
        want = {k: (v.clone() if torch.is_tensor(v) else {kk: vv.clone() for kk, vv in v.items()}) for k, v in want.items()}
        got = our_model(x)
    n = 0
    for t, v in want.items():
        for tt, vv in ([(None, v)] if torch.is_tensor(v) else v.items()):
            g = got[t] if tt is None else got[t][tt]
            assert g.shape == vv.shape and _rel(g, vv) < 1e-3, (t, tt, _rel(g, vv))
            n += 1
    mtt_amd.ops.clear_pack_cache()
    return n


def test_taskprompter_through_the_references_own_get_model(tmp_path, monkeypatch):
    """pascal_vitBp16_taskprompter.yml's model keys (backbone TaskPrompter_vitB, head conv, ctr) at a small input and narrow decoder."""
    import mtt_amd
    monkeypatch.setenv("TORCH_HOME", str(tmp_path))
    assert torch.hub.get_dir().startswith(str(tmp_path))
    _write_flax_vit(mtt_amd.checkpoints.cached_pretrained_path("vit_base_patch16_384"), 768, 12, 12, 3, seed=11)
    ED = ref_import.easydict()
    size = (64, 96)

    def make_p():
        p = ED(dict(model="TaskPrompter", backbone="TaskPrompter_vitB", head="conv", embed_dim=44, final_embed_dim=52, prompt_len=1,
                    chan_nheads=4, use_ctr=True))
        p.TASKS = ED(dict(NAMES=["semseg", "human_parts", "sal", "normals", "edge"],
                          NUM_OUTPUT=dict(semseg=21, human_parts=7, sal=2, normals=3, edge=1)))
        p.TRAIN = ED(dict(SCALE=size))
        return p
    try:
        torch.manual_seed(0)
        ref_model = _import_factory("TaskPrompter", {}).get_model(make_p())
        assert type(ref_model).__module__ == "models.taskprompter_wrapper"
        factory = _import_factory("TaskPrompter", {"models.transformers.taskprompter": mtt_amd.taskprompter,
                                                   "models.taskprompter_wrapper": mtt_amd.taskprompter})
        assert "/reference/" in factory.__file__                      # the reference's file, not factory.py of this package
        p2 = make_p()
        p2.mtt_prec = "x3"
        our_model = factory.get_model(p2)
        assert isinstance(our_model, mtt_amd.taskprompter.TaskPrompterWrapper)
        assert isinstance(our_model.heads["semseg"], mtt_amd.taskprompter.ConvHead)
        assert p2.backbone_channels == 52 and p2.spatial_dim == [[4, 6]] * 4          # what the factory writes into p is untouched
    finally:
        _cleanup()
    assert _compare(ref_model, our_model, size, monkeypatch) == 5


def test_invpt_through_the_references_own_get_model(tmp_path, monkeypatch):
    """InvPT/utils/common_config.py: backbone 'vitL' (ViT-L/16, 24 blocks, pretrained=True) + TransformerDecoder + MLPHead, 2 tasks, 64x64."""
    import mtt_amd
    monkeypatch.setenv("TORCH_HOME", str(tmp_path))
    _write_flax_vit(mtt_amd.checkpoints.cached_pretrained_path("vit_large_patch16_384"), 1024, 24, 16, 3, seed=12)
    ED = ref_import.easydict()
    size = (64, 64)

    def make_p():
        p = ED(dict(model="TransformerNet", backbone="vitL", head="mlp", embed_dim=32, PRED_OUT_NUM_CONSTANT=8, mtt_resolution_downsample_rate=2))
        p.TASKS = ED(dict(NAMES=["semseg", "depth"], NUM_OUTPUT=dict(semseg=40, depth=1)))
        p.TRAIN = ED(dict(SCALE=size))
        return p
    try:
        torch.manual_seed(0)
        ref_model = _import_factory("InvPT", {}).get_model(make_p())
        assert type(ref_model).__module__ == "models.transformer_net"
        factory = _import_factory("InvPT", {"models.transformers.vit": mtt_amd.invpt, "models.transformers.transformer_decoder": mtt_amd.invpt,
                                            "models.transformer_net": mtt_amd.invpt})
        p2 = make_p()
        p2.mtt_prec = "x3"
        our_model = factory.get_model(p2)
        assert isinstance(our_model, mtt_amd.invpt.TransformerNet)
    finally:
        _cleanup()
    assert _compare(ref_model, our_model, size, monkeypatch) == 4         # 2 heads + 2 inter_preds
