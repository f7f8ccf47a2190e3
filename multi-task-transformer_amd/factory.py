"""Host-side mirror of the reference's model factory (TaskPrompter/utils/common_config.py:17-90,
InvPT/utils/common_config.py:12-51): `get_model(p)` returns the drop-in nn.Module for a reference-style
config object `p`.  `p` may be the reference's EasyDict or the `AttrDict` below — the models only use
attribute access, item access and `keys()`.

Extra (non-reference) key: `p.mtt_prec` in {'x3f', 'bf16', 'x3'} selects the arithmetic mode.  Default `ops.DEFAULT_PREC` = 'x3f': the mode whose
forward matches the reference's fp32 forward within north_star's 1e-3 per task head (bf16 does not: 1.5e-2).
"""
import torch

from .ops import DEFAULT_PREC


class AttrDict(dict):
    """Minimal EasyDict-compatible container (attribute access == item access, nested dicts wrapped)."""

    def __init__(self, d=None, **kw):
        super().__init__()
        for k, v in dict(d or {}, **kw).items():
            self[k] = v

    def __setitem__(self, k, v):
        if isinstance(v, dict) and not isinstance(v, AttrDict):
            v = AttrDict(v)
        super().__setitem__(k, v)

    __setattr__ = __setitem__

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e


PASCAL_NUM_OUTPUT = dict(semseg=21, human_parts=7, sal=2, normals=3, edge=1, depth=1)
TASK_ORDER = ["semseg", "depth", "human_parts", "sal", "normals", "edge"]   # TaskPrompter/utils/config.py:30-87


def make_p(tasks, img_size, model="TaskPrompter", backbone="TaskPrompter_vitL", head="conv", embed_dim=300,
           final_embed_dim=350, prompt_len=1, chan_nheads=1, use_ctr=True, num_output=None, prec=None, **extra):
    """Reference-style config for synthetic runs (what create_config would produce from a YAML)."""
    nout = dict(PASCAL_NUM_OUTPUT, **(num_output or {}))
    p = AttrDict(model=model, backbone=backbone, head=head, embed_dim=embed_dim, final_embed_dim=final_embed_dim,
                 prompt_len=prompt_len, chan_nheads=chan_nheads, use_ctr=use_ctr, mtt_prec=prec or DEFAULT_PREC)
    p.TASKS = AttrDict(NAMES=list(tasks), NUM_OUTPUT=AttrDict({t: nout[t] for t in tasks}))
    p.TRAIN = AttrDict(SCALE=tuple(img_size))
    for k, v in extra.items():
        p[k] = v
    return p


def get_backbone(p):
    """TaskPrompter/utils/common_config.py:17-46 (pretrained weights are never downloaded: load a checkpoint)."""
    from . import taskprompter as tp
    if p['backbone'] == 'TaskPrompter_vitL':
        backbone = tp.taskprompter_vit_large_patch16_384(p=p, pretrained=False, drop_path_rate=p.get('drop_path_rate', 0.15),
                                                         img_size=p.TRAIN.SCALE)
    elif p['backbone'] == 'TaskPrompter_vitB':
        backbone = tp.taskprompter_vit_base_patch16_384(p=p, pretrained=False, drop_path_rate=p.get('drop_path_rate', 0.15),
                                                        img_size=p.TRAIN.SCALE)
    elif p['backbone'] == 'TaskPrompter_swinB' or isinstance(p['backbone'], dict):
        # TaskPrompter/utils/common_config.py:34-41; a dict(patch_size=, window_size=, embed_dim=, depths=, num_heads=) builds other sizes
        from . import taskprompter_swin as sw
        kw = dict(p['backbone']) if isinstance(p['backbone'], dict) else dict(patch_size=4, window_size=12, embed_dim=128,
                                                                              depths=(2, 2, 18, 2), num_heads=(4, 8, 16, 32))
        e = kw['embed_dim']
        p.backbone_channels = [2 * e, 4 * e, 8 * e, 8 * e]
        img_h, img_w = p.TRAIN.SCALE
        p.ori_spatial_dim = [[img_h // st, img_w // st] for st in (8, 16, 32, 32)]
        backbone = sw.taskprompter_create_swin_transformer('swin', pretrained=False, p=p, drop_path_rate=p.get('drop_path_rate', 0.15),
                                                           img_size=p.TRAIN.SCALE, **kw)
        return backbone, p.final_embed_dim
    elif isinstance(p['backbone'], (tuple, list)):       # (embed_dim, depth, heads, select_list): miniature / ViT-S variants
        C, depth, nH, select = p['backbone']
        backbone = tp._create_task_prompter('custom', p=p, select_list=list(select), patch_size=16, embed_dim=C, depth=depth,
                                            num_heads=nH, chan_nheads=p.chan_nheads,
                                            drop_path_rate=p.get('drop_path_rate', 0.0), img_size=p.TRAIN.SCALE)
    else:
        raise NotImplementedError(p['backbone'])
    p.backbone_channels = p.final_embed_dim
    p.spatial_dim = [[p.TRAIN.SCALE[0] // 16, p.TRAIN.SCALE[1] // 16] for _ in range(4)]
    return backbone, p.final_embed_dim


def get_head(p, backbone_channels, task):
    from . import taskprompter as tp
    if task == '3ddet':
        raise NotImplementedError('FCOS3D head depends on mmcv/mmdet3d (out of scope, SURVEY.md §2 #17)')
    if p['head'] == 'conv':
        return tp.ConvHead(backbone_channels, p.TASKS.NUM_OUTPUT[task])
    if p['head'] == 'deconv':
        return tp.DEConvHead(backbone_channels, p.TASKS.NUM_OUTPUT[task])
    raise NotImplementedError(p['head'])


def get_invpt_backbone(p):
    """InvPT/utils/common_config.py:12-25."""
    from . import invpt
    prec = p.get('mtt_prec', DEFAULT_PREC)
    if p['backbone'] == 'vitL':
        backbone = invpt.vit_large_patch16_384(pretrained=False, drop_path_rate=p.get('drop_path_rate', 0.15), img_size=p.TRAIN.SCALE, prec=prec)
        C = 1024
    elif isinstance(p['backbone'], (tuple, list)):
        C, depth, nH, select = p['backbone']
        backbone = invpt._create_vision_transformer('custom', select_list=list(select), patch_size=16, embed_dim=C, depth=depth, num_heads=nH,
                                                    drop_path_rate=p.get('drop_path_rate', 0.0), img_size=p.TRAIN.SCALE, prec=prec)
    else:
        raise NotImplementedError(p['backbone'])
    p.backbone_channels = [C for _ in range(4)]
    p.spatial_dim = [[p.TRAIN.SCALE[0] // 16, p.TRAIN.SCALE[1] // 16] for _ in range(4)]
    p.final_embed_dim = p.embed_dim + p.PRED_OUT_NUM_CONSTANT
    return backbone, p.backbone_channels


def get_model(p):
    """TaskPrompter/utils/common_config.py:76-90 and InvPT/utils/common_config.py:39-51."""
    if p['model'] == 'TransformerNet':
        from . import invpt
        backbone, ch = get_invpt_backbone(p)
        heads = torch.nn.ModuleDict({task: invpt.MLPHead(p.final_embed_dim, p.TASKS.NUM_OUTPUT[task]) for task in p.TASKS.NAMES})
        return invpt.TransformerNet(p, backbone, ch, heads)
    if p['model'] == 'TaskPrompter':
        from . import taskprompter as tp
        backbone, ch = get_backbone(p)
        heads = torch.nn.ModuleDict({task: get_head(p, ch, task) for task in p.TASKS.NAMES})
        return tp.TaskPrompterWrapper(p, backbone, heads)
    raise NotImplementedError('Unknown model {}'.format(p['model']))
