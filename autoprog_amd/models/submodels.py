"""`model_variant('volo_h{H}_l{L}')` factory (reference models/submodels.py:9-40) used by the
AutoProg trainers.  The as-shipped reference asserts variant == 'volo_h12_l18' before parsing
(SURVEY.md 0.1-1); this implements the intended behaviour: any even H, any L."""
from ..prog.helpers import split_depth
from .registry import register_model
from .volo import VOLO, default_cfgs


def parse_variant(variant):
    parts = variant.split("_")
    if len(parts) != 3 or not parts[1].startswith("h") or not parts[2].startswith("l"):
        raise ValueError("variant must look like 'volo_h12_l18', got %r" % (variant,))
    return parts[0], int(parts[1].lstrip("h")), int(parts[2].lstrip("l"))


@register_model
def model_variant(variant="", pretrained=False, **kwargs):
    """a general model with variants of any size: heads H (embed 16H/32H), total depth L"""
    family, h, l = parse_variant(variant)
    if family == "deit":
        from .deit import deit_variant
        return deit_variant(h, l, **kwargs)
    if family != "volo":
        raise ValueError("unknown family %r" % family)
    assert h % 2 == 0, "h must be divisible by 2"
    if l <= 2:
        print("Warning: layer too small, set to 2")
    model = VOLO(split_depth(l), embed_dims=[h * 16, h * 32, h * 32, h * 32], num_heads=[h // 2, h, h, h],
                 mlp_ratios=[3, 3, 3, 3], downsamples=[True, False, False, False],
                 outlook_attention=[True, False, False, False], post_layers=["ca", "ca"], **kwargs)
    model.default_cfg = default_cfgs["volo"]
    return model
