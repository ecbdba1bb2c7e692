"""Layer-index maps used when depth grows (reference prog/helpers.py:254-262) and the
active-layer mask that `VOLO.set_sample_config` derives from them (models/volo.py:598-616).

The reference keeps the "which layers are identity" decision implicit in two places that
disagree after stage 0 (search uses min_layer_num = next depth, extraction uses base_layer =
previous depth).  Here it is ONE explicit object, `ActiveLayerMask`,
shared by the elastic forward and by any extraction code."""
from .progressive import make_divisible


def new_idx(idx, prev_l, new_l):
    """prog/helpers.py:254-258: index of the source layer (in a prev_l-deep stage) whose
    weights initialise destination layer `idx` of the grown new_l-deep stage."""
    reps = new_l // prev_l
    single = prev_l - new_l % prev_l
    cand = idx * prev_l // (reps * prev_l)
    if cand < single:
        return cand
    return (idx + single) * prev_l // (reps * prev_l + prev_l)


def get_new_layer_idx(prev_l, new_l):
    """prog/helpers.py:261-262: destination layers that are clones of their predecessor."""
    return [i for i in range(new_l) if new_idx(i, prev_l, new_l) == new_idx(i - 1, prev_l, new_l)]


def split_depth(l):
    """total depth -> [outlooker, transformer, 0, 0] (models/submodels.py:19-25)."""
    if l > 2:
        l0 = make_divisible(l * 0.23, 2)
        return [l0, l - l0, 0, 0]
    return [1, 1, 0, 0]


class ActiveLayerMask:
    """per-stage sets of identity (skipped) layer indices for an elastic-depth config
    {layer_num, min_layer_num, max_layer_num} (models/volo.py:598-609)."""

    def __init__(self, layer_num, min_layer_num, max_layer_num):
        def parts(l):
            l0 = make_divisible(l * 0.23, 2)
            return [l0, l - l0, 0, 0]
        cur, lo, hi = parts(layer_num), parts(min_layer_num), parts(max_layer_num)
        self.config = dict(layer_num=layer_num, min_layer_num=min_layer_num, max_layer_num=max_layer_num)
        self.skip = []
        for s in range(4):
            fresh = get_new_layer_idx(lo[s], hi[s]) if hi[s] > 0 else []
            grown = cur[s] - lo[s]
            self.skip.append(frozenset(fresh if grown == 0 else fresh[:-grown]))

    def is_identity(self, stage, layer):
        return layer in self.skip[stage]

    def kept_layers(self, stage, depth):
        return [i for i in range(depth) if i not in self.skip[stage]]
