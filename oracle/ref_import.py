"""TEST INFRASTRUCTURE ONLY — import the UNMODIFIED reference model files on CPU.

Only `tests/`, `tests/golden/make_golden.py` and `bench.py`'s cpu_baseline leg may use this.
It never copies reference sources: it puts `/root/reference/<sub-project>` on `sys.path`
(the reference uses absolute imports `models.…`, `utils.utils`) together with the stand-in
`timm` / `easydict` packages in `oracle/_refshim` (timm==0.5.4 and easydict are pinned by the
reference but absent from this image; SURVEY.md §8c lists the 12 symbols).

`/root/reference` exists only in the build container, never on the GPU box: everything that
must travel is generated here by `tests/golden/make_golden.py` and committed as fixtures.
"""
import importlib
import os
import sys
from types import SimpleNamespace

REFERENCE_ROOT = os.environ.get("MTT_REFERENCE_ROOT", "/root/reference")
_SHIM = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_refshim")
_SUBPROJECT = {"TP": "TaskPrompter", "TPS": "TaskPrompter", "IP": "InvPT"}


def reference_available():
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "TaskPrompter", "models"))


def _purge():
    for name in list(sys.modules):
        root = name.split(".")[0]
        if root in ("models", "utils", "configs", "losses"):
            del sys.modules[name]


def load_reference(which):
    """Return the reference's own classes for sub-project `which` in {"TP", "IP"}.

    TP -> taskprompter.py symbols + TaskPrompterWrapper; IP -> vit.py, transformer_decoder.py,
    invpt.py, transformer_net.py symbols.  Both sub-projects own top-level packages called
    `models`/`utils`, so `sys.modules` is purged before and after the import.
    """
    if not reference_available():
        raise RuntimeError(f"reference not found under {REFERENCE_ROOT}")
    root = os.path.join(REFERENCE_ROOT, _SUBPROJECT[which])
    _purge()
    added = [p for p in (_SHIM, root) if p not in sys.path]
    for p in added:
        sys.path.insert(0, p)
    try:
        ns = SimpleNamespace()
        if which == "TP":
            tp = importlib.import_module("models.transformers.taskprompter")
            wr = importlib.import_module("models.taskprompter_wrapper")
            ns.module = tp
            ns.TaskPrompter = tp.TaskPrompter
            ns.Attention = tp.Attention
            ns.Block = tp.Block
            ns.ConvHead = tp.ConvHead
            ns.DEConvHead = tp.DEConvHead
            ns.vit_large = tp.taskprompter_vit_large_patch16_384
            ns.vit_base = tp.taskprompter_vit_base_patch16_384
            ns.create = tp._create_task_prompter
            ns.TaskPrompterWrapper = wr.TaskPrompterWrapper
        elif which == "TPS":
            sw = importlib.import_module("models.transformers.taskprompter_swin")
            tp = importlib.import_module("models.transformers.taskprompter")
            wr = importlib.import_module("models.taskprompter_wrapper")
            ns.module = sw
            ns.TaskPrompterSwin = sw.TaskPrompterSwin
            ns.create = sw.taskprompter_create_swin_transformer
            ns.swin_base = sw.taskprompter_swin_base_patch4_window12_384
            ns.ConvHead = tp.ConvHead
            ns.DEConvHead = tp.DEConvHead
            ns.TaskPrompterWrapper = wr.TaskPrompterWrapper
        else:
            vit = importlib.import_module("models.transformers.vit")
            dec = importlib.import_module("models.transformers.transformer_decoder")
            inv = importlib.import_module("models.transformers.invpt")
            net = importlib.import_module("models.transformer_net")
            ns.vit = vit
            ns.VisionTransformer = vit.VisionTransformer
            ns.create_vit = vit._create_vision_transformer
            ns.vit_large = vit.vit_large_patch16_384
            ns.TransformerDecoder = dec.TransformerDecoder
            ns.MLPHead = dec.MLPHead
            ns.ConvBlock = dec.ConvBlock
            ns.InvPT = inv.InvPT
            ns.TransformerNet = net.TransformerNet
        return ns
    finally:
        for p in added:
            if p in sys.path:
                sys.path.remove(p)
        _purge()


def easydict():
    if _SHIM not in sys.path:
        sys.path.insert(0, _SHIM)
        try:
            return importlib.import_module("easydict").EasyDict
        finally:
            sys.path.remove(_SHIM)
    return importlib.import_module("easydict").EasyDict
