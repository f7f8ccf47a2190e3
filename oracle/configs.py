"""TEST INFRASTRUCTURE ONLY — named model configurations as plain dicts.

Each entry restates the model-shaping keys of a reference YAML (cited) or of a BASELINE.json
config that the reference's own parser can express.  Task ORDER is the order
`parse_task_dictionary` emits (TaskPrompter/utils/config.py:30-87), not YAML order:
semseg, depth, human_parts, sal, normals, edge.
"""

PASCAL5 = (("semseg", 21), ("human_parts", 7), ("sal", 2), ("normals", 3), ("edge", 1))
PASCAL6 = (("semseg", 21), ("depth", 1), ("human_parts", 7), ("sal", 2), ("normals", 3), ("edge", 1))
NYUD4 = (("semseg", 40), ("depth", 1), ("normals", 3), ("edge", 1))
NYUD2 = (("semseg", 40), ("depth", 1))
CS2 = (("semseg", 19), ("depth", 1))

VIT = {  # (embed_dim, depth, heads, select_list)
    "nano": (64, 4, 1, (1, 2, 3)),       # test-only: checkpoint-import fixtures
    "tiny": (128, 4, 2, (1, 2, 3)),      # test-only miniature, same code path
    "tiny6": (128, 6, 2, (2, 4, 5)),     # test-only: blocks 0 and 2 are NOT taps (their logit side channels have no consumer, as 20 of ViT-L's 24)
    "small": (384, 12, 6, (3, 6, 9)),
    "base": (768, 12, 12, (3, 6, 9)),    # taskprompter.py:683
    "large": (1024, 24, 16, (6, 12, 18)),  # taskprompter.py:675, vit.py:560
}


def taskprompter(name):
    """TaskPrompter configs.  Keys: backbone, img_size, tasks, embed_dim(tar), final_embed_dim(F),
    chan_nheads, use_ctr, prompt_len, head ('conv'|'deconv')."""
    t = {
        # TaskPrompter/configs/pascal/pascal_vitLp16_taskprompter.yml:26-33  (+depth = 6 tasks)
        "ns6": dict(backbone="large", img_size=(512, 512), tasks=PASCAL6, embed_dim=300, final_embed_dim=350,
                    chan_nheads=1, use_ctr=True, prompt_len=1, head="conv"),
        "ns5": dict(backbone="large", img_size=(512, 512), tasks=PASCAL5, embed_dim=300, final_embed_dim=350,
                    chan_nheads=1, use_ctr=True, prompt_len=1, head="conv"),
        # TaskPrompter/configs/pascal/pascal_vitBp16_taskprompter.yml (cfg2)
        "cfg2": dict(backbone="base", img_size=(512, 512), tasks=PASCAL5, embed_dim=780, final_embed_dim=1024,
                     chan_nheads=16, use_ctr=True, prompt_len=1, head="conv"),
        # TaskPrompter/configs/nyud/nyud_vitLp16_taskprompter.yml (cfg3)
        "cfg3": dict(backbone="large", img_size=(448, 576), tasks=NYUD4, embed_dim=768, final_embed_dim=768,
                     chan_nheads=16, use_ctr=False, prompt_len=1, head="conv"),
        # BASELINE.json configs[4] as resolved by SURVEY §0 (ViT-L, DEConvHead, 2 dense tasks)
        "cfg5": dict(backbone="large", img_size=(1024, 2048), tasks=CS2, embed_dim=300, final_embed_dim=350,
                     chan_nheads=1, use_ctr=True, prompt_len=1, head="deconv"),
        # miniatures for fast CPU parity (same code paths: odd dims, windows, ctr, deconv)
        "mini_ctr": dict(backbone="tiny", img_size=(64, 96), tasks=PASCAL6, embed_dim=44, final_embed_dim=52,
                         chan_nheads=1, use_ctr=True, prompt_len=1, head="conv"),
        # the wrapper's `dd_label_map_size` branch (taskprompter_wrapper.py:17-27): mini_ctr with predictions resized to 40 x 56 instead of
        # the 64 x 96 input (fixture tests/golden/mini_ctr_dd.npz from the unmodified reference; same state-dict contract as mini_ctr)
        "mini_ctr_dd": dict(backbone="tiny", img_size=(64, 96), tasks=PASCAL6, embed_dim=44, final_embed_dim=52,
                            chan_nheads=1, use_ctr=True, prompt_len=1, head="conv", dd_label_map_size=(40, 56), contract_of="mini_ctr"),
        # six blocks, taps after blocks 2 / 4 / 5 + the last: the product skips the channel-logit pass and the side-channel gradients of the
        # non-tap blocks (round 6); no reference fixture: product vs oracle
        "mini_skip": dict(backbone="tiny6", img_size=(64, 96), tasks=NYUD4, embed_dim=48, final_embed_dim=40,
                          chan_nheads=4, use_ctr=True, prompt_len=1, head="conv"),
        "mini_win": dict(backbone="tiny", img_size=(64, 96), tasks=NYUD4, embed_dim=48, final_embed_dim=40,
                         chan_nheads=4, use_ctr=False, prompt_len=1, head="conv"),
        "mini_deconv": dict(backbone="tiny", img_size=(64, 64), tasks=CS2, embed_dim=30, final_embed_dim=36,
                            chan_nheads=1, use_ctr=True, prompt_len=1, head="deconv"),
        # decoder widths that are multiples of 32 (as 352 = pad8(350), 1024, 768 of the BASELINE configs): the x3f mode then runs the
        # fea_fuse 3x3 convs and the taps-first head GEMM on split planes.  No reference fixture: product vs oracle only.
        "mini_p32": dict(backbone="tiny", img_size=(64, 96), tasks=NYUD4, embed_dim=48, final_embed_dim=64,
                         chan_nheads=1, use_ctr=True, prompt_len=1, head="conv"),
    }[name]
    return dict(t, name=name, model="TaskPrompter")


def swin(name):
    """TaskPrompter-Swin configs (taskprompter_swin.py:545-700; cs_swinB_taskprompter.yml:13-15,31-40).  Keys: patch, window, embed
    (stage-0 dim), depths, heads, img_size, img_ds_ratio, tasks, level_embed_dim, final_embed_dim, chan_embed_dim, chan_nheads,
    prompt_len, head.  backbone_channels / strides follow common_config.py:36-38 ([2, 4, 8, 8] x embed, strides [8, 16, 32, 32])."""
    t = {
        # TaskPrompter/configs/cityscapes3d/cs_swinB_taskprompter.yml without the 3ddet task (FCOS3D head needs mmdet3d: absent)
        "cs_swinB": dict(patch=4, window=12, embed=128, depths=(2, 2, 18, 2), heads=(4, 8, 16, 32), img_size=(1024, 2048), img_ds_ratio=0.75,
                         tasks=CS2, level_embed_dim=256, final_embed_dim=450, chan_embed_dim=256, chan_nheads=1, prompt_len=1, head="deconv"),
        # miniatures: same code paths (shifted windows with masks, 4 stages with patch merging, head dim 32 like Swin-B, DEConv heads)
        "mini_swin": dict(patch=4, window=4, embed=32, depths=(2, 2, 2, 2), heads=(1, 2, 4, 8), img_size=(128, 192), img_ds_ratio=1.0,
                          tasks=CS2, level_embed_dim=24, final_embed_dim=40, chan_embed_dim=16, chan_nheads=4, prompt_len=1, head="deconv"),
        # level width 32 (padded concatenation 64 = two 32-deep K steps) and final width 64: in x3f the task features' GEMMs / 3x3 convs and
        # (with the row thresholds lowered by the test) the stage Linears run on split planes.  No reference fixture: product vs oracle.
        "mini_swin_sp": dict(patch=4, window=4, embed=32, depths=(2, 2, 2, 2), heads=(1, 2, 4, 8), img_size=(128, 192), img_ds_ratio=1.0,
                             tasks=CS2, level_embed_dim=32, final_embed_dim=64, chan_embed_dim=16, chan_nheads=4, prompt_len=1, head="deconv"),
        # window 5 does not divide the 48 x 72 ... 6 x 9 grids (zero padding after norm1, also with the shift), 0.75 input resize,
        # 3 tasks, ConvHeads
        "mini_swin_pad": dict(patch=4, window=5, embed=32, depths=(2, 2, 2, 2), heads=(1, 2, 4, 8), img_size=(256, 384), img_ds_ratio=0.75,
                              tasks=(("semseg", 19), ("depth", 1), ("normals", 3)), level_embed_dim=16, final_embed_dim=24,
                              chan_embed_dim=64, chan_nheads=1, prompt_len=1, head="conv"),
    }[name]
    return dict(t, name=name, model="TaskPrompterSwin")


def invpt(name):
    """InvPT configs.  Keys: backbone, img_size, tasks, embed_dim, pred_const, mtt_down."""
    t = {
        # InvPT/configs/pascal/pascal_vitLp16.yml:24-29 (cfg4; +depth = 6 tasks)
        "cfg4_6": dict(backbone="large", img_size=(512, 512), tasks=PASCAL6, embed_dim=512, pred_const=64, mtt_down=2),
        "cfg4_5": dict(backbone="large", img_size=(512, 512), tasks=PASCAL5, embed_dim=512, pred_const=64, mtt_down=2),
        # BASELINE.json configs[0]: ViT-S built through the parametric constructor
        "cfg1": dict(backbone="small", img_size=(256, 256), tasks=NYUD2, embed_dim=512, pred_const=64, mtt_down=2),
        "mini": dict(backbone="tiny", img_size=(128, 64), tasks=NYUD2, embed_dim=32, pred_const=8, mtt_down=2),
        # every channel / head dim a multiple of 8 (64/32/16, heads 32/16/8) like the published configs: training-path tests
        "mini8": dict(backbone="tiny", img_size=(128, 64), tasks=NYUD2, embed_dim=56, pred_const=8, mtt_down=2),
    }[name]
    return dict(t, name=name, model="TransformerNet")


def to_p(cfg, attrdict):
    """Build the reference-style `p` (the EasyDict main.py passes to get_model) from a config dict."""
    names = [n for n, _ in cfg["tasks"]]
    nout = {n: c for n, c in cfg["tasks"]}
    p = attrdict()
    p.TASKS = attrdict(NAMES=names, NUM_OUTPUT=attrdict(nout))
    H, W = cfg["img_size"]
    p.TRAIN = attrdict(SCALE=(H, W))
    p.spatial_dim = [[H // 16, W // 16] for _ in range(4)]
    if cfg["model"] == "TaskPrompterSwin":
        e = cfg["embed"]
        p.backbone_channels = [2 * e, 4 * e, 8 * e, 8 * e]                    # common_config.py:36
        p.ori_spatial_dim = [[H // st, W // st] for st in (8, 16, 32, 32)]    # common_config.py:37-39
        p.img_ds_ratio = cfg["img_ds_ratio"]
        p.fea_ds_ratio = 1
        p.level_embed_dim = cfg["level_embed_dim"]
        p.final_embed_dim = cfg["final_embed_dim"]
        p.chan_embed_dim = cfg["chan_embed_dim"]
        p.chan_nheads = cfg["chan_nheads"]
        p.prompt_len = cfg["prompt_len"]
        p.head = cfg["head"]
        return p
    if cfg["model"] == "TaskPrompter":
        p.embed_dim = cfg["embed_dim"]
        p.final_embed_dim = cfg["final_embed_dim"]
        p.prompt_len = cfg["prompt_len"]
        p.chan_nheads = cfg["chan_nheads"]
        p.use_ctr = cfg["use_ctr"]
        p.head = cfg["head"]
        p.backbone_channels = cfg["final_embed_dim"]
    else:
        C = VIT[cfg["backbone"]][0]
        p.embed_dim = cfg["embed_dim"]
        p.PRED_OUT_NUM_CONSTANT = cfg["pred_const"]
        p.mtt_resolution_downsample_rate = cfg["mtt_down"]
        p.backbone_channels = [C] * 4
        p.final_embed_dim = cfg["embed_dim"] + cfg["pred_const"]
        p.head = "mlp"
    return p
