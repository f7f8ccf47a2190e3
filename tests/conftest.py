import json
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


HOST_THREADS = min(16, os.cpu_count() or 1)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the CPU oracle / emulator legs: a fixed, modest thread count.  The GPU box has 256 hardware threads and torch's default (all of
    # them) THRASHES on these workloads — measured in round 2: the NS-6 oracle forward took 220 s on 256 threads vs 5 s on 8.
    torch.set_num_threads(HOST_THREADS)
    # A/B of kernel variants under the model-level parity tests (a test-harness switch, not a library one): every mtt_gemm call that
    # leaves its descriptor at AUTO gets this variant, e.g. MTT_TEST_GEMM_VARIANT=1 = the register-staged general kernel everywhere
    if os.environ.get("MTT_TEST_GEMM_VARIANT"):
        import mtt_amd
        mtt_amd.ops.GEMM_VARIANT = int(os.environ["MTT_TEST_GEMM_VARIANT"])
    # the multiple-of-32 channel pitch (ops.pad8) starts at 160 channels: no miniature has a map that wide, so the same suite runs a second
    # time with MTT_TEST_PITCH32_FROM=33 (every ragged channel count of the miniatures then takes the wide pitch: 44 -> 64, 52 -> 64 ...)
    if os.environ.get("MTT_TEST_PITCH32_FROM"):
        import mtt_amd
        mtt_amd.ops.PITCH32_FROM = int(os.environ["MTT_TEST_PITCH32_FROM"])


def has_gpu():
    return torch.cuda.is_available()


def load_golden(name):
    meta = json.load(open(os.path.join(GOLDEN, f"{name}.json")))
    arrs = np.load(os.path.join(GOLDEN, f"{name}.npz"))
    return meta, arrs


def build_product_model(cfg, prec, device="cpu", drop_path_rate=0.0):
    """Product nn.Module for an oracle/configs.py config dict."""
    import mtt_amd
    from oracle import configs
    if cfg["model"] == "TaskPrompterSwin":
        p = mtt_amd.factory.make_p([t for t, _ in cfg["tasks"]], cfg["img_size"],
                                   backbone=dict(patch_size=cfg["patch"], window_size=cfg["window"], embed_dim=cfg["embed"],
                                                 depths=tuple(cfg["depths"]), num_heads=tuple(cfg["heads"])),
                                   head=cfg["head"], final_embed_dim=cfg["final_embed_dim"], chan_nheads=cfg["chan_nheads"],
                                   num_output=dict(cfg["tasks"]), prec=prec, drop_path_rate=drop_path_rate, img_ds_ratio=cfg["img_ds_ratio"],
                                   level_embed_dim=cfg["level_embed_dim"], chan_embed_dim=cfg["chan_embed_dim"], prompt_len=cfg["prompt_len"])
        return mtt_amd.factory.get_model(p).to(device)
    C, depth, nH, sel = configs.VIT[cfg["backbone"]]
    if cfg["model"] == "TransformerNet":
        p = mtt_amd.factory.make_p([t for t, _ in cfg["tasks"]], cfg["img_size"], model="TransformerNet", backbone=(C, depth, nH, sel),
                                   head="mlp", embed_dim=cfg["embed_dim"], num_output=dict(cfg["tasks"]), prec=prec,
                                   PRED_OUT_NUM_CONSTANT=cfg["pred_const"], mtt_resolution_downsample_rate=cfg["mtt_down"],
                                   drop_path_rate=drop_path_rate)
        return mtt_amd.factory.get_model(p).to(device)
    extra = dict(dd_label_map_size=tuple(cfg["dd_label_map_size"])) if cfg.get("dd_label_map_size") is not None else {}
    p = mtt_amd.factory.make_p([t for t, _ in cfg["tasks"]], cfg["img_size"], backbone=(C, depth, nH, sel), head=cfg["head"],
                               embed_dim=cfg["embed_dim"], final_embed_dim=cfg["final_embed_dim"], chan_nheads=cfg["chan_nheads"],
                               use_ctr=cfg["use_ctr"], num_output=dict(cfg["tasks"]), prec=prec, drop_path_rate=drop_path_rate, **extra)
    return mtt_amd.factory.get_model(p).to(device)


@pytest.fixture
def emulated(monkeypatch):
    """Route the product's C-ABI calls to the CPU emulator (host-wiring tests without a GPU)."""
    import mtt_amd
    from oracle import abi_emul
    monkeypatch.setattr(mtt_amd.ops, "call", abi_emul.call)
    mtt_amd.ops.clear_pack_cache()
    yield
    mtt_amd.ops.clear_pack_cache()
