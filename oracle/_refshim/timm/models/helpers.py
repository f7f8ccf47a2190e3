"""build_model_with_cfg / named_apply / adapt_input_conv with timm 0.5.4 semantics (no download)."""
import torch.nn as nn


def named_apply(fn, module, name='', depth_first=True, include_root=False):
    if not depth_first and include_root:
        fn(module=module, name=name)
    for child_name, child in module.named_children():
        child_name = '.'.join((name, child_name)) if name else child_name
        named_apply(fn=fn, module=child, name=child_name, depth_first=depth_first, include_root=True)
    if depth_first and include_root:
        fn(module=module, name=name)
    return module


def adapt_input_conv(in_chans, conv_weight):
    assert in_chans == conv_weight.shape[1]
    return conv_weight


def overlay_external_default_cfg(default_cfg, kwargs):
    kwargs.pop('external_default_cfg', None)


def build_model_with_cfg(model_cls, variant, pretrained, default_cfg, model_cfg=None, feature_cfg=None,
                         pretrained_strict=True, pretrained_filter_fn=None, pretrained_custom_load=False,
                         kwargs_filter=None, **kwargs):
    # timm injects num_classes / in_chans / img_size defaults from default_cfg when present;
    # img_size only when the cfg says fixed_input_size (taskprompter.py:28-37 sets it True).
    if 'num_classes' in default_cfg:
        kwargs.setdefault('num_classes', default_cfg['num_classes'])
    input_size = default_cfg.get('input_size', None)
    if input_size is not None:
        kwargs.setdefault('in_chans', input_size[0])
        if default_cfg.get('fixed_input_size', False):
            kwargs.setdefault('img_size', input_size[-2:])
    if kwargs_filter:
        for k in kwargs_filter:
            kwargs.pop(k, None)
    model = model_cls(**kwargs) if model_cfg is None else model_cls(cfg=model_cfg, **kwargs)
    model.default_cfg = default_cfg
    if pretrained:
        # timm 0.5.4: pretrained_custom_load -> load_custom_pretrained(model): download_cached_file(url) keeps the file under
        # <torch.hub.get_dir()>/checkpoints/<basename of the URL> and re-uses it when present, then model.load_pretrained(file).
        # Offline stand-in: the cached file only.
        import os
        from urllib.parse import urlparse
        import torch
        if not pretrained_custom_load:
            raise RuntimeError('torch (.pth) pretrained weights cannot be downloaded offline; call with pretrained=False')
        cached = os.path.join(torch.hub.get_dir(), 'checkpoints', os.path.basename(urlparse(default_cfg['url']).path))
        if not os.path.isfile(cached):
            raise RuntimeError('pretrained weights cannot be downloaded offline and %s is not cached' % cached)
        model.load_pretrained(cached)
    return model
