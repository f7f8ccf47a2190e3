"""multi-task-transformer_amd — MI355X (gfx950) native hot path of prismformore/Multi-Task-Transformer.

The directory name carries a hyphen (it mirrors the reference repository's name), so import it through
`importlib.import_module("multi-task-transformer_amd")` or the `mtt_amd` alias module at the repo root.
"""
from . import _lib, ops, autograd_path, factory, losses, det_losses, optim, checkpoints, iou3d, taskprompter_swin, graphs  # noqa: F401
from .taskprompter import (ConvHead, DEConvHead, TaskPrompter, TaskPrompterWrapper,  # noqa: F401
                           taskprompter_vit_base_patch16_384, taskprompter_vit_large_patch16_384)
from .invpt import (MLPHead, TransformerDecoder, TransformerNet, VisionTransformer, vit_large_patch16_384)  # noqa: F401,E402
from .taskprompter_swin import TaskPrompterSwin, taskprompter_swin_base_patch4_window12_384  # noqa: F401,E402
