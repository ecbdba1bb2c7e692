"""Stage transitions: autoprog_amd.prog.growth against fingerprints of the reference's own
load_slice_clone_ema / load_super (tools/gen_golden_growth.py ran them on /root/reference with the same deterministic
weights).  Pure index / slicing arithmetic: bit exact."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _detfill import fill_state_dict, fingerprint      # noqa: E402

GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "growth.npz"))


def _sd(variant):
    from autoprog_amd.models import create_model
    m = create_model("model_variant", variant=variant, num_classes=16, img_size=64, stem_hidden_dim=16)
    return m, {k: v.detach().clone() for k, v in m.state_dict().items()}


def _check(case, sd):
    keys = [k[len(case) + 1:] for k in GOLD.files if k.startswith(case + "/")]
    assert sorted(keys) == sorted(sd.keys())
    for k in keys:
        want = GOLD[case + "/" + k]
        got = np.asarray(fingerprint(sd[k]), dtype=np.float64)
        assert np.array_equal(got, want), (case, k, got[:3], want[:3])


@pytest.mark.parametrize("case,src_v,dst_v", [("deeper", "volo_h2_l4", "volo_h2_l6"), ("wider_deeper", "volo_h2_l4", "volo_h4_l7"),
                                              ("wider", "volo_h2_l6", "volo_h4_l6")])
def test_grow_clone_ema_matches_reference(case, src_v, dst_v):
    from autoprog_amd.prog.growth import grow_clone_ema
    _, src = _sd(src_v)
    dst_model, dst = _sd(dst_v)
    emas = [fill_state_dict(src, i + 1) for i in range(4)]
    # reference call shape: load_slice_clone_ema(model, prev_ema_list[3], prev_ema_list): the "model" source is EMA 3
    grown = grow_clone_ema(fill_state_dict(dst, 9), emas[3], emas)
    _check(case, grown)
    dst_model.load_state_dict(grown)                     # shapes and keys fit the destination network


@pytest.mark.parametrize("case,sup_v,sub_v,base", [("sub_l4_of_l7", "volo_h2_l7", "volo_h2_l4", 4), ("sub_l5_of_l7", "volo_h2_l7", "volo_h2_l5", 4)])
def test_extract_subnet_matches_reference(case, sup_v, sub_v, base):
    from autoprog_amd.prog.growth import extract_subnet
    _, sup = _sd(sup_v)
    sub_model, sub = _sd(sub_v)
    got = extract_subnet(fill_state_dict(sub, 8), fill_state_dict(sup, 3), base)
    _check(case, got)
    sub_model.load_state_dict(got)


def test_subnet_layer_map_agrees_with_elastic_mask():
    """the layers the elastic forward keeps for {layer_num, min, max} are the layers extraction copies"""
    from autoprog_amd.prog.growth import subnet_layer_map
    from autoprog_amd.prog.helpers import ActiveLayerMask, split_depth
    for lmin, lmax in [(4, 7), (6, 9), (12, 18)]:
        for l in range(lmin, lmax + 1):
            mask = ActiveLayerMask(l, lmin, lmax)
            sub, sup = split_depth(l), split_depth(lmax)
            m = subnet_layer_map({0: sub[0], 2: sub[1]}, {0: sup[0], 2: sup[1]}, lmin)
            assert m[0] == mask.kept_layers(0, sup[0]) and m[2] == mask.kept_layers(1, sup[1]), (lmin, lmax, l)


def test_search_helpers_match_reference():
    """no_repeats / get_divisor / sample_configs against the reference's own functions (cut out of main_prog.py and run by
    tools/gen_golden_growth.py); search_space / converge_speed: behaviour checks (inline code in the reference, unpinned)"""
    import json
    import random
    from autoprog_amd.prog import search as S
    rec = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "search.json")))
    for n, f, want in rec["get_divisor"]:
        assert S.get_divisor(n, f) == want, (n, f)
    for v, want in rec["no_repeats"]:
        assert S.no_repeats(v) == want
    for seed, cfgs in rec["sample_configs"]:
        random.seed(seed)
        for want in cfgs:
            got = S.sample_configs([12, 15, 18], [160, 192, 224], mode="random")
            assert [got[0], got[1], got[2]] == want
    assert list(S.sample_configs([9, 12], [128, 160], mode="smallest")) == rec["sample_smallest"]
    # stage 0: first / middle / last; later stages: a window from the current point, depth one step ahead
    r_list, h_list, l_list = [128, 160, 192, 224], [12, 12, 12, 12], [9, 12, 15, 18]
    assert S.search_space(0, r_list, h_list, l_list, 128, 12, 9) == ([128, 192, 224], [12], [9, 15, 18])
    assert S.search_space(1, r_list, h_list, l_list, 160, 12, 12) == ([160, 192], [12], [15, 18])
    assert S.search_space(2, r_list, h_list, l_list, 224, 12, 18) == ([224], [12], [18])
    # loss = 2 * t^-0.5 exactly -> w = 0.5 and every candidate scores the same; a candidate below the curve wins
    t = {"a": 1.0, "b": 2.0, "c": 4.0}
    loss = {k: 2.0 * v ** -0.5 for k, v in t.items()}
    w, scores, order = S.converge_speed(loss, t)
    assert abs(w - 0.5) < 1e-6 and max(scores.values()) - min(scores.values()) < 1e-6
    loss["b"] *= 0.9
    assert S.converge_speed(loss, t)[2][0] == "b"
