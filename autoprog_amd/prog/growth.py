"""Stage transitions of a progressive run on reference-format state dicts.

  grow_clone_ema      a bigger network (more layers and/or up to 2x wider) initialised from the previous stage's model
                      and its EMA copies -- the semantics of the reference's load_slice_clone_ema
                      (prog/helpers.py:613-749; call site main_prog.py:1382)
  extract_subnet      a sub-network of a trained supernet: which supernet layer feeds sub-network layer i
                      (load_super, prog/helpers.py:752-902, equal widths)

Both are pure index / slicing arithmetic on host tensors, executed once per stage; the elastic forward itself never
copies weights (models/volo.py ActiveLayerMask).  Rules restated from the reference:
  * a destination layer i of a stage that became deeper takes its weights from source layer new_idx(i, old, new); the
    stages with blocks are network.0/2/3/4 (network.1 is the down-sampling conv)
  * when the destination is wider, every weight matrix is cut out of the 2x2 tiling
        [[model, ema0], [ema1, ema2]]         (rows = output channels, columns = input channels)
    and Linear weights are divided by (new in-features / old in-features) so that activations keep their scale; the
    down-sampling conv (network.1) divides the same way, other convolutions do not; packed projections are tiled per
    group (qkv: 3 groups, kv: 2 groups) so that each of q/k/v grows on its own
  * vectors (biases, norm weights, pos_embed, cls_token) are cut out of [model, ema0]
  * BatchNorm running statistics are NOT carried over (the reference leaves them at their initial values)
"""
import torch

from .helpers import get_new_layer_idx, new_idx
from .progressive import make_divisible

_BLOCK_STAGES = (0, 2, 3, 4)


def stage_depths(state_dict, prefix="network."):
    """number of blocks per network stage, read off the keys (stage 1 = down-sampling conv has none)"""
    depth = {}
    for k in state_dict:
        if k.startswith(prefix):
            parts = k[len(prefix):].split(".")
            if len(parts) > 2 and parts[0].isdigit() and parts[1].isdigit():
                s = int(parts[0])
                depth[s] = max(depth.get(s, 0), int(parts[1]) + 1)
    return depth


def _source_key(key, dst_depth, src_depth):
    parts = key.split(".")
    if parts[0] == "network" and len(parts) > 3 and parts[1].isdigit() and parts[2].isdigit():
        s = int(parts[1])
        if s in _BLOCK_STAGES and dst_depth.get(s, 0) > src_depth.get(s, 0) > 0:
            parts[2] = str(new_idx(int(parts[2]), src_depth[s], dst_depth[s]))
    return ".".join(parts)


def _tile_crop(blocks, out_c, in_c, groups):
    """[[a, b], [c, d]] tiled per group, cropped to [out_c, in_c, ...]"""
    a, b, c, d = blocks
    so = a.shape[0] // groups
    tail = a.shape[2:]
    v = lambda t: t.reshape(groups, so, t.shape[1], *tail)
    top = torch.cat([v(a), v(b)], dim=2)
    bot = torch.cat([v(c), v(d)], dim=2)
    full = torch.cat([top, bot], dim=1)
    return full[:, :out_c // groups, :in_c].reshape(out_c, in_c, *tail)


def grow_clone_ema(dst_state_dict, src_state_dict, ema_state_dicts):
    """returns a new state dict with dst's keys / shapes: every tensor that has a source is overwritten following the
    rules above, everything else (e.g. BatchNorm statistics) keeps dst's value.  `ema_state_dicts`: at least 3."""
    if len(ema_state_dicts) < 3:
        raise ValueError("grow_clone_ema needs the model and three EMA state dicts")
    dst_depth, src_depth = stage_depths(dst_state_dict), stage_depths(src_state_dict)
    e0, e1, e2 = ema_state_dicts[:3]
    out = {}
    for key, dst in dst_state_dict.items():
        leaf = key.rsplit(".", 1)[-1]
        skey = _source_key(key, dst_depth, src_depth)
        if skey not in src_state_dict or leaf in ("running_mean", "running_var", "num_batches_tracked"):
            out[key] = dst.clone()
            continue
        a = src_state_dict[skey]
        if dst.ndim >= 2 and leaf == "weight":
            out_c, in_c = dst.shape[0], dst.shape[1]
            if out_c > 2 * a.shape[0] or in_c > 2 * a.shape[1]:
                raise ValueError("%s: growth by more than 2x is not defined" % key)
            owner = key[:-len(".weight")]
            groups = 3 if owner.endswith(".qkv") else (2 if owner.endswith(".kv") else 1)
            w = _tile_crop((a, e0[skey], e1[skey], e2[skey]), out_c, in_c, groups)
            is_conv = dst.ndim == 4
            if (not is_conv) or key.startswith("network.1."):
                w = w / (in_c / a.shape[1])
            out[key] = w.to(dst.dtype)
        elif dst.ndim >= 1 and dst.dtype.is_floating_point:
            # vectors and pos_embed / cls_token grow along their LAST axis; packed biases per group
            n = dst.shape[-1]
            if n > 2 * a.shape[-1]:
                raise ValueError("%s: growth by more than 2x is not defined" % key)
            owner = key.rsplit(".", 1)[0]
            groups = 3 if (owner.endswith(".qkv") and leaf == "bias") else (2 if (owner.endswith(".kv") and leaf == "bias") else 1)
            lead = a.shape[:-1]
            av = a.reshape(*lead, groups, a.shape[-1] // groups)
            bv = e0[skey].reshape(*lead, groups, a.shape[-1] // groups)
            out[key] = torch.cat([av, bv], dim=-1)[..., :n // groups].reshape(dst.shape).to(dst.dtype)
        else:
            out[key] = dst.clone()
    return out


def subnet_layer_map(sub_depth, super_depth, base_layer):
    """for every block stage: the supernet layer that feeds each sub-network layer (load_super's index rule).
    `base_layer`: total depth of the smallest network of the supernet (an int, split 23 % / 77 % like split_depth)."""
    if base_layer > 2:
        l0 = make_divisible(base_layer * 0.23, 2)
        base = {0: l0, 1: 0, 2: base_layer - l0, 3: 0, 4: 0}
    else:
        base = {0: 1, 1: 0, 2: 1, 3: 0, 4: 0}
    mapping = {}
    for s in _BLOCK_STAGES:
        n_sub, n_sup = sub_depth.get(s, 0), super_depth.get(s, 0)
        if n_sub == 0:
            continue
        if n_sub == n_sup:
            mapping[s] = list(range(n_sub))
        elif n_sub > n_sup:
            mapping[s] = [new_idx(i, n_sup, n_sub) for i in range(n_sub)]
        else:
            fresh = get_new_layer_idx(base[s], n_sup)
            extra = n_sub - base[s]
            skipped = fresh[:-extra] if extra > 0 else fresh
            kept = [i for i in range(n_sup) if i not in skipped]
            if len(kept) != n_sub:
                raise ValueError("stage %d: %d layers cannot be extracted from %d with base %d" % (s, n_sub, n_sup, base[s]))
            mapping[s] = kept
    return mapping


def extract_subnet(sub_state_dict, super_state_dict, base_layer):
    """state dict with sub's keys whose block weights come from the supernet layers subnet_layer_map() selects
    (equal widths; BatchNorm statistics keep sub's values as in the reference)."""
    sub_depth, super_depth = stage_depths(sub_state_dict), stage_depths(super_state_dict)
    mapping = subnet_layer_map(sub_depth, super_depth, base_layer)
    out = {}
    for key, dst in sub_state_dict.items():
        parts = key.split(".")
        skey = key
        if parts[0] == "network" and len(parts) > 3 and parts[1].isdigit() and parts[2].isdigit() and int(parts[1]) in mapping:
            parts[2] = str(mapping[int(parts[1])][int(parts[2])])
            skey = ".".join(parts)
        leaf = parts[-1]
        if skey in super_state_dict and leaf not in ("running_mean", "running_var", "num_batches_tracked") \
                and super_state_dict[skey].shape == dst.shape:
            out[key] = super_state_dict[skey].clone()
        else:
            out[key] = dst.clone()
    return out
