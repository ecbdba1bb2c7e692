"""One parameter set for every stage of a progressive run (north_star: "elastic depth ...
realised as kernel launch-time shape parameters rather than weight copies").

The reference builds a NEW network at every grow epoch and fills it from the previous one (create_stage_model_and_optimizer,
main_prog.py:1300-1430: load_slice_clone_ema for the model, load_slice_clone for the EMA copies, prog/helpers.py:349-478,613-749).
Here the network, its Adam moments and its EMA copies are allocated ONCE for the deepest stage (the supernet of the schedule);
a stage is an ActiveLayerMask over it.  The active sets are nested (a deeper stage only un-skips layers), so a transition
  * leaves the physical slots of the surviving layers where they are,
  * (re)initialises every slot that is active in the new stage from the slot that holds the reference's source layer
    new_idx(j, prev, new) -- model weights from the LAST EMA copy, EMA copy i from EMA copy i -- exactly the tensors the
    reference would copy into its freshly built network,
  * restarts the optimizer (moments, step count) and the BatchNorm running statistics, as the reference does.
`export_state_dict` / `import_state_dict` translate between the physical slots and the reference's checkpoint format (layers
numbered 0..l-1 per stage, prog/checkpoint_saver.py:110-130), so checkpoints stay interchangeable with it.
"""
import re

import torch

from .helpers import ActiveLayerMask, new_idx

_STAGE_OF_NETWORK = {0: 0, 2: 1, 3: 2, 4: 3}          # network.{0,2,3,4} hold blocks; network.1 is the down-sampling conv
_BLOCK_KEY = re.compile(r"^network\.(\d+)\.(\d+)\.(.+)$")


def physical_layers(mask, max_depths):
    """{network stage index: [physical slot of logical layer 0, 1, ...]} of an ActiveLayerMask over a supernet whose stages
    hold max_depths[s] blocks (s = 0..3)"""
    return {net: mask.kept_layers(s, max_depths[s]) for net, s in _STAGE_OF_NETWORK.items() if max_depths[s] > 0}


def _max_depths(keys):
    depth = [0, 0, 0, 0]
    for k in keys:
        m = _BLOCK_KEY.match(k)
        if m and int(m.group(1)) in _STAGE_OF_NETWORK:
            s = _STAGE_OF_NETWORK[int(m.group(1))]
            depth[s] = max(depth[s], int(m.group(2)) + 1)
    return depth


def export_state_dict(super_sd, mask):
    """reference-format state dict of the ACTIVE sub-network: blocks renumbered 0..l-1 per stage, everything else unchanged"""
    phys = physical_layers(mask, _max_depths(super_sd))
    out = {}
    for k, v in super_sd.items():
        m = _BLOCK_KEY.match(k)
        if m and int(m.group(1)) in phys:
            net, slot = int(m.group(1)), int(m.group(2))
            if slot not in phys[net]:
                continue
            k = "network.%d.%d.%s" % (net, phys[net].index(slot), m.group(3))
        out[k] = v
    return out


def import_state_dict(sub_sd, mask, super_keys):
    """inverse of export_state_dict: {supernet key: tensor} for every key of the active sub-network (load with strict=False)"""
    phys = physical_layers(mask, _max_depths(super_keys))
    out = {}
    for k, v in sub_sd.items():
        m = _BLOCK_KEY.match(k)
        if m and int(m.group(1)) in phys:
            net, j = int(m.group(1)), int(m.group(2))
            k = "network.%d.%d.%s" % (net, phys[net][j], m.group(3))
        out[k] = v
    return out


def growth_sources(names, old_mask, new_mask):
    """{destination parameter name: source parameter name} of a stage transition old_mask -> new_mask on one supernet: the layer in
    logical position j of the new stage is initialised from the layer in logical position new_idx(j, prev, new) of the old one
    (prog/helpers.py:254-258,356-361); parameters outside the block stages map to themselves; slots that stay inactive are absent."""
    depths = _max_depths(names)
    old_p, new_p = physical_layers(old_mask, depths), physical_layers(new_mask, depths)
    out = {}
    for k in names:
        m = _BLOCK_KEY.match(k)
        if not (m and int(m.group(1)) in new_p):
            out[k] = k
            continue
        net, slot = int(m.group(1)), int(m.group(2))
        if slot not in new_p[net]:
            continue
        j = new_p[net].index(slot)
        prev, new = len(old_p[net]), len(new_p[net])
        # growth: the reference's source layer; same depth or a SHALLOWER stage (the sub-network chosen by a search out of its
        # supernet, load='super', main_prog.py:830-837): the active sets are nested, so every kept layer is its own source
        src_slot = old_p[net][new_idx(j, prev, new)] if new > prev else slot
        out[k] = "network.%d.%d.%s" % (net, src_slot, m.group(3))
    return out


def make_mask(layer_num, min_layer_num, max_layer_num):
    return ActiveLayerMask(layer_num, min_layer_num, max_layer_num)
