"""Minimal model registry with timm 0.4.5 `create_model` semantics: kwargs whose
value is None are dropped, bn_* are popped, drop_connect_rate aliases drop_path_rate."""
_ENTRYPOINTS = {}


def register_model(fn):
    _ENTRYPOINTS[fn.__name__] = fn
    return fn


def is_model(name):
    return name in _ENTRYPOINTS


def list_models():
    return sorted(_ENTRYPOINTS)


def create_model(model_name, pretrained=False, checkpoint_path="", scriptable=None, exportable=None, no_jit=None, **kwargs):
    for k in ("bn_tf", "bn_momentum", "bn_eps"):
        kwargs.pop(k, None)
    dcr = kwargs.pop("drop_connect_rate", None)
    if dcr is not None and kwargs.get("drop_path_rate", None) is None:
        kwargs["drop_path_rate"] = dcr
    kwargs = {k: v for k, v in kwargs.items() if v is not None}
    if model_name not in _ENTRYPOINTS:
        raise RuntimeError("Unknown model (%s)" % model_name)
    if pretrained:
        raise RuntimeError("pretrained weights are not downloadable in this environment")
    model = _ENTRYPOINTS[model_name](pretrained=False, **kwargs)
    if checkpoint_path:
        import torch
        state = torch.load(checkpoint_path, map_location="cpu")
        model.load_state_dict(state.get("state_dict", state))
    return model
