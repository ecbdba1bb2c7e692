"""Import the read-only AutoProg reference (/root/reference) in THIS container only.

timm / tlt / apex are not installed here, so a minimal stand-in for the six timm
symbols the hot-path modules import is placed in ``sys.modules`` first (SURVEY.md
section 8(c) row O1).  None of the stubbed symbols performs arithmetic on the parity
path when ``drop_path_rate == 0`` and weights come from a state dict.

This file is tooling for generating golden vectors (tools/gen_golden.py) and for
here-only cross-checks.  It is never imported by the product, by bench.py or by
``-m gpu`` tests: /root/reference does not exist on the GPU box.
"""
import os
import sys
import types

REF_ROOT = os.environ.get("AUTOPROG_REFERENCE", "/root/reference")


def reference_available() -> bool:
    return os.path.isfile(os.path.join(REF_ROOT, "models", "volo.py"))


def _install_timm_stub():
    import numpy as np
    import torch
    import torch.nn as nn

    if not hasattr(np, "int"):          # reference models/volo.py:327 uses np.int
        np.int = int

    timm = types.ModuleType("timm")
    data = types.ModuleType("timm.data")
    data.IMAGENET_DEFAULT_MEAN = (0.485, 0.456, 0.406)
    data.IMAGENET_DEFAULT_STD = (0.229, 0.224, 0.225)
    models = types.ModuleType("timm.models")
    layers = types.ModuleType("timm.models.layers")
    registry = types.ModuleType("timm.models.registry")
    vit = types.ModuleType("timm.models.vision_transformer")
    utils = types.ModuleType("timm.utils")

    class DropPath(nn.Module):          # timm 0.4.5 semantics, SURVEY.md A.1
        def __init__(self, drop_prob=None):
            super().__init__()
            self.drop_prob = drop_prob

        def forward(self, x):
            if not self.drop_prob or not self.training:
                return x
            keep = 1.0 - self.drop_prob
            shape = (x.shape[0],) + (1,) * (x.ndim - 1)
            mask = (keep + torch.rand(shape, dtype=x.dtype, device=x.device)).floor_()
            return x.div(keep) * mask

    def to_2tuple(v):
        return tuple(v) if isinstance(v, (tuple, list)) else (v, v)

    def trunc_normal_(t, mean=0.0, std=1.0, a=-2.0, b=2.0):
        return nn.init.trunc_normal_(t, mean=mean, std=std, a=a, b=b)

    _registry = {}

    def register_model(fn):
        _registry[fn.__name__] = fn
        return fn

    layers.DropPath, layers.to_2tuple, layers.trunc_normal_ = DropPath, to_2tuple, trunc_normal_
    registry.register_model = register_model
    registry._stub_registry = _registry
    vit.VisionTransformer = type("VisionTransformer", (nn.Module,), {})
    vit._cfg = lambda **kw: dict(kw)
    utils.unwrap_model = lambda m: m.module if hasattr(m, "module") else m
    timm.data, timm.models, timm.utils = data, models, utils
    models.layers, models.registry, models.vision_transformer = layers, registry, vit
    for name, mod in [("timm", timm), ("timm.data", data), ("timm.models", models),
                      ("timm.models.layers", layers), ("timm.models.registry", registry),
                      ("timm.models.vision_transformer", vit), ("timm.utils", utils)]:
        sys.modules.setdefault(name, mod)


class _RefNamespace:
    pass


def load_reference():
    """Return a namespace with the reference modules (volo, submodels, cross_entropy,
    helpers, progressive).  Raises RuntimeError when the reference tree is absent."""
    if not reference_available():
        raise RuntimeError("reference tree not present at %s" % REF_ROOT)
    _install_timm_stub()
    # The reference uses top-level package names (models, loss, prog) that collide with
    # nothing in this repo (ours live under autoprog_amd.*), so a plain path insert works.
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    import importlib
    ns = _RefNamespace()
    ns.volo = importlib.import_module("models.volo")
    ns.submodels = importlib.import_module("models.submodels")
    ns.cross_entropy = importlib.import_module("loss.cross_entropy")
    ns.helpers = importlib.import_module("prog.helpers")
    ns.progressive = importlib.import_module("prog.progressive")
    return ns
