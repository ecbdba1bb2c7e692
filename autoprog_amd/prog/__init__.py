"""Host-side integer bookkeeping of AutoProg (schedule, layer-index maps)."""
from .progressive import make_divisible, progressive_schedule  # noqa: F401
from .helpers import new_idx, get_new_layer_idx, ActiveLayerMask  # noqa: F401

from .growth import extract_subnet, grow_clone_ema, stage_depths, subnet_layer_map  # noqa: E402,F401
from .search import converge_speed, get_divisor, no_repeats, sample_configs, search_space  # noqa: E402,F401
