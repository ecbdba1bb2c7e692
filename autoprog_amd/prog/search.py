"""Host-side pieces of the AutoProg growth search: which candidates a stage searches, how a
sub-network config is sampled per step, how the candidates are ranked, and the batch-split rule.

Pinned against the reference's own functions (tools/gen_golden_growth.py extracts them from main_prog.py with `ast` and runs
them here): `no_repeats`, `get_divisor`, `sample_configs`.  `search_space` and `converge_speed` restate code that is inline
in the reference's 270-line `main()` / `auto_grow` bodies (main_prog.py:793-803, 1780-1806) and cannot be called in
isolation: parity unpinned, same scipy call.
"""
import random as _random


def no_repeats(values):
    """order-preserving de-duplication (main_prog.py:2064-2069)"""
    out = []
    for v in values:
        if v not in out:
            out.append(v)
    return out


def get_divisor(number, factor):
    """smallest divisor of `number` that is > number*factor (main_prog.py:2057-2061): the number of micro-batches a
    stage needs when its activation footprint is `factor` of the largest stage's"""
    for i in range(int(number * factor) + 1, number + 1):
        if number % i == 0:
            return i
    return number


def sample_configs(l_list, r_list, mode="random", rng=_random):
    """one elastic sub-network config per training step of the supernet (main_prog.py:1824-1837)"""
    if mode == "random":
        layer_num, input_size = rng.choice(l_list), rng.choice(r_list)
    elif mode == "smallest":
        layer_num, input_size = l_list[0], r_list[0]
    else:
        raise NotImplementedError(mode)
    config = {"min_layer_num": l_list[0], "max_layer_num": l_list[-1], "layer_num": layer_num, "input_size": input_size,
              "token_label_size": input_size // 16}
    return config, l_list.index(layer_num), r_list.index(input_size)


def search_space(stage, r_list, h_list, l_list, current_r, current_h, current_l):
    """candidate resolutions / heads / depths of the search that opens stage `stage` (main_prog.py:793-803): stage 0 looks at
    first / middle / last of the whole schedule, later stages at up to 2 resolutions and 3 depths from the current point on
    (the depth window starts one step ahead when it can)"""
    rs, hs, ls = no_repeats(r_list), no_repeats(h_list), no_repeats(l_list)
    if stage > 0:
        r_s, h_s, l_s = rs.index(current_r), hs.index(current_h), ls.index(current_l)
        if l_s < len(ls) - 1:
            l_s += 1
        return rs[r_s:min(r_s + 2, len(rs))], hs[h_s:min(h_s + 3, len(hs))], ls[l_s:min(l_s + 3, len(ls))]
    return [rs[0], rs[len(rs) // 2], rs[-1]], hs, [ls[0], ls[len(ls) // 2], ls[-1]]


def converge_speed(mean_loss, step_time):
    """rank candidates by loss * time^w (main_prog.py:1793-1806): w = max(-a1, 0) of the power law loss = a2 * time^a1 fitted
    over the candidates (scipy.optimize.curve_fit); smaller is better.  Returns (w, scores, names sorted best first)."""
    from scipy.optimize import curve_fit
    names = list(mean_loss)
    x = [step_time[n] for n in names]
    y = [mean_loss[n] for n in names]
    para, _ = curve_fit(lambda t, a1, a2: a2 * t ** a1, x, y)
    w = max(-para[0], 0)
    scores = {n: mean_loss[n] * step_time[n] ** w for n in names}
    return w, scores, sorted(scores, key=scores.get)
