"""AutoProg stage schedule (reference prog/progressive.py:4-40), bit-exact integer/float
bookkeeping on the host.  `progressive_schedule` accepts the reference's argparse namespace."""
import numpy as np


def make_divisible(v, divisor=8, min_value=None, round_limit=0.9):
    """prog/progressive.py:34-40: round to the nearest multiple of `divisor` (at least
    `min_value`), bumping one step up if that lost more than 10 %."""
    lowest = min_value or divisor
    rounded = max(lowest, int(v + divisor / 2) // divisor * divisor)
    return rounded + divisor if rounded < round_limit * v else rounded


def _ramp(lo, n):
    return np.linspace(lo, 1.0, n)


def progressive_schedule(args, r_max=224, h_max=12, l_max=18):
    """prog/progressive.py:4-31 -> (epochs, r, h, l, aa, dp, re, resize) per stage."""
    n = args.num_stages
    epochs = [int(i) for i in np.linspace(0, args.epochs, n + 1) // 1][:-1]
    res = [make_divisible(i, 32) for i in _ramp(args.r_scale, n) * r_max]
    heads = [make_divisible(i, 2) for i in _ramp(args.h_scale, n) * h_max]      # even head counts only
    depth = [make_divisible(i, 1) for i in _ramp(args.l_scale, n) * l_max]
    if not (isinstance(args.aa, str) and args.aa.startswith("rand")):
        raise ValueError("progressive_schedule needs a rand-m* auto-augment spec")
    m_top = float(args.aa.split("-")[1].lstrip("m"))
    mags = [round(max(0.0, i)) for i in _ramp(args.aa_scale, n) * m_top]
    aa = ["rand-m{}-mstd0.5-inc1".format(m) if m > 0 else "" for m in mags]
    dp = [max(0.0, i) for i in _ramp(args.dp_scale, n) * args.drop_path]
    re = [max(0.0, i) for i in _ramp(args.re_scale, n) * args.reprob]
    lo = _ramp(args.resize_scale[0], n) * args.scale[0]
    hi = _ramp(args.resize_scale[1], n) * args.scale[1]
    resize = [[max(0.0, a), max(0.0, b)] for a, b in zip(lo, hi)]
    return epochs, res, heads, depth, aa, dp, re, resize
