"""Seed-reproducible state dict with the statistics of the reference's own initialisation, shared by
tools/gen_golden.py (reference side) and the GPU loss-curve test, so the realistic-init fixture needs no
stored weights.

Rule per tensor = what `VOLO.__init__` + `_init_weights` leave behind (/root/reference models/volo.py:559-566,
:555-557, :541): Linear weights, pos_embed and cls_token trunc-normal(std .02, cut at +-2), Linear biases 0,
LayerNorm (1, 0); Conv2d keeps torch's default (kaiming-uniform a=sqrt(5): U(+-1/sqrt(fan_in)) for weight and
bias); BatchNorm (1, 0), running stats (0, 1), num_batches_tracked 0.  Values come from numpy's RandomState
(bit-identical on every host), drawn in state-dict order."""
import numpy as np
import torch


def init_state_dict(ref_sd, seed):
    """ref_sd: name -> tensor (shapes/dtypes only are used); returns name -> fp32/int64 tensor"""
    rng = np.random.RandomState(seed)
    out = {}
    for name, t in ref_sd.items():
        shape = tuple(t.shape)
        leaf = name.rsplit(".", 1)[-1]
        if not t.dtype.is_floating_point:                       # num_batches_tracked
            out[name] = torch.zeros(shape, dtype=t.dtype)
        elif leaf == "running_mean":
            out[name] = torch.zeros(shape)
        elif leaf == "running_var":
            out[name] = torch.ones(shape)
        elif name in ("pos_embed", "cls_token", "dist_token"):
            out[name] = torch.from_numpy(np.clip(rng.normal(0.0, 0.02, shape), -2.0, 2.0).astype(np.float32))
        elif len(shape) == 4:                                   # Conv2d weight
            bound = 1.0 / np.sqrt(shape[1] * shape[2] * shape[3])
            out[name] = torch.from_numpy(rng.uniform(-bound, bound, shape).astype(np.float32))
        elif len(shape) == 2:                                   # Linear weight
            out[name] = torch.from_numpy(np.clip(rng.normal(0.0, 0.02, shape), -2.0, 2.0).astype(np.float32))
        elif leaf == "bias" and name.replace(".bias", ".weight") in ref_sd and ref_sd[name.replace(".bias", ".weight")].dim() == 4:
            w = ref_sd[name.replace(".bias", ".weight")]
            bound = 1.0 / np.sqrt(w.shape[1] * w.shape[2] * w.shape[3])
            out[name] = torch.from_numpy(rng.uniform(-bound, bound, shape).astype(np.float32))
        elif leaf == "weight":                                  # LayerNorm / BatchNorm scale
            out[name] = torch.ones(shape)
        else:                                                   # biases
            out[name] = torch.zeros(shape)
    return out
