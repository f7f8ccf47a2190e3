"""TEST INFRASTRUCTURE ONLY — construct the UNMODIFIED reference nn.Modules for a config dict."""
import torch

from . import configs, ref_import


def randomize_norm_state(model, seed=123):
    """Give BN running stats / affine and LN affine non-trivial values so folding bugs show."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for m in model.modules():
            if isinstance(m, torch.nn.modules.batchnorm._BatchNorm):
                m.running_mean.copy_(torch.randn(m.running_mean.shape, generator=g) * 0.1)
                m.running_var.copy_(torch.rand(m.running_var.shape, generator=g) * 0.5 + 0.75)
                m.weight.copy_(torch.rand(m.weight.shape, generator=g) * 0.5 + 0.75)
                m.bias.copy_(torch.randn(m.bias.shape, generator=g) * 0.1)
            elif isinstance(m, torch.nn.LayerNorm):
                m.weight.copy_(torch.rand(m.weight.shape, generator=g) * 0.5 + 0.75)
                m.bias.copy_(torch.randn(m.bias.shape, generator=g) * 0.1)
            elif isinstance(m, (torch.nn.Linear, torch.nn.Conv2d, torch.nn.ConvTranspose2d)) and m.bias is not None:
                m.bias.copy_(torch.randn(m.bias.shape, generator=g) * 0.02)


def build_reference(cfg, seed=0, drop_path_rate=0.0, randomize=True):
    """Reference model for `cfg` (oracle/configs.py), weights under torch.manual_seed(seed)."""
    ED = ref_import.easydict()
    p = configs.to_p(cfg, ED)
    torch.manual_seed(seed)
    if cfg["model"] == "TaskPrompterSwin":
        ns = ref_import.load_reference("TPS")
        backbone = ns.create("swin_base_patch4_window12_384", pretrained=False, p=p, patch_size=cfg["patch"], window_size=cfg["window"],
                             embed_dim=cfg["embed"], depths=tuple(cfg["depths"]), num_heads=tuple(cfg["heads"]),
                             drop_path_rate=drop_path_rate, img_size=tuple(cfg["img_size"]))
        Head = ns.ConvHead if cfg["head"] == "conv" else ns.DEConvHead
        heads = torch.nn.ModuleDict({t: Head(p.final_embed_dim, n) for t, n in cfg["tasks"]})
        model = ns.TaskPrompterWrapper(p, backbone, heads)
        if randomize:
            randomize_norm_state(model)
        return model, p
    C, depth, nH, select = configs.VIT[cfg["backbone"]]
    if cfg["model"] == "TaskPrompter":
        ns = ref_import.load_reference("TP")
        backbone = ns.create("vit_large_patch16_384", pretrained=False, p=p, select_list=list(select), patch_size=16,
                             embed_dim=C, depth=depth, num_heads=nH, chan_nheads=p.chan_nheads,
                             drop_path_rate=drop_path_rate, img_size=tuple(cfg["img_size"]))
        Head = ns.ConvHead if cfg["head"] == "conv" else ns.DEConvHead
        heads = torch.nn.ModuleDict({t: Head(p.final_embed_dim, n) for t, n in cfg["tasks"]})
        model = ns.TaskPrompterWrapper(p, backbone, heads)
    else:
        ns = ref_import.load_reference("IP")
        backbone = ns.create_vit("vit_large_patch16_384", pretrained=False, select_list=list(select), patch_size=16,
                                 embed_dim=C, depth=depth, num_heads=nH, drop_path_rate=drop_path_rate,
                                 img_size=tuple(cfg["img_size"]))
        heads = torch.nn.ModuleDict({t: ns.MLPHead(p.final_embed_dim, n) for t, n in cfg["tasks"]})
        model = ns.TransformerNet(p, backbone, p.backbone_channels, heads)
    if randomize:
        randomize_norm_state(model)
    return model, p
