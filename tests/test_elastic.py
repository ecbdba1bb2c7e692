"""Stage transitions on ONE supernet (prog/elastic.py): the physical-slot arithmetic against the state-dict functions of
prog/growth.py, which are pinned bit-exact against the reference's load_slice_clone_ema / load_super (tests/test_growth.py)."""
import torch

from autoprog_amd.prog import elastic, growth
from autoprog_amd.prog.helpers import ActiveLayerMask, split_depth


def _fake_sd(l, tag, width=4):
    """reference-format state dict of a volo_h*_l{l}-shaped network (block stages network.0 / network.2), distinct values per tensor"""
    d0, d1 = split_depth(l)[:2]
    sd = {"pos_embed": torch.full((1, 2, 2, width), 0.5 + tag), "patch_embed.proj.weight": torch.full((width, 3, 2, 2), 0.25 + tag)}
    for net, depth in ((0, d0), (2, d1)):
        for i in range(depth):
            sd["network.%d.%d.mlp.fc1.weight" % (net, i)] = torch.full((width, width), 100.0 * net + i + tag)
            sd["network.%d.%d.norm1.bias" % (net, i)] = torch.full((width,), 100.0 * net + i + 0.5 + tag)
    sd["network.1.proj.weight"] = torch.full((width, width, 2, 2), 7.0 + tag)
    return sd


def test_export_import_round_trip_and_nesting():
    sup = _fake_sd(18, 0.0)
    prev = None
    for l in (9, 12, 15, 18):
        mask = ActiveLayerMask(l, 9, 18)
        sub = elastic.export_state_dict(sup, mask)
        assert growth.stage_depths(sub) == {0: split_depth(l)[0], 2: split_depth(l)[1]}
        back = elastic.import_state_dict(sub, mask, sup.keys())
        assert all(torch.equal(back[k], sup[k]) for k in back) and set(back) <= set(sup)
        slots = elastic.physical_layers(mask, [4, 14, 0, 0])
        if prev is not None:                       # a deeper stage only un-skips layers
            assert all(set(prev[n]) <= set(slots[n]) for n in slots)
        prev = slots
    # supernet(config l) == what the reference's load_super extracts when base_layer == min_layer_num
    for l in (9, 12, 15):
        sub = elastic.export_state_dict(sup, ActiveLayerMask(l, 9, 18))
        ref = growth.extract_subnet(_fake_sd(l, 9.0), sup, base_layer=9)
        assert all(torch.equal(sub[k], ref[k]) for k in ref)


def test_growth_sources_match_grow_clone_ema():
    """in-place transition l_prev -> l_new on the supernet == grow_clone_ema on the extracted state dicts (equal widths)"""
    names = list(_fake_sd(18, 0.0))
    for l_prev, l_new in ((9, 12), (12, 15), (15, 18), (9, 18), (12, 12)):
        old_mask, new_mask = ActiveLayerMask(l_prev, 9, 18), ActiveLayerMask(l_new, 9, 18)
        ema = [_fake_sd(18, 1000.0 * (i + 1)) for i in range(4)]                 # supernet-shaped EMA copies, all different
        src_of = elastic.growth_sources(names, old_mask, new_mask)
        # in-place result for the model (from the LAST EMA copy) and for EMA copy 1
        model_new = {k: ema[3][src_of[k]].clone() for k in names if k in src_of}
        ema1_new = {k: ema[1][src_of[k]].clone() for k in names if k in src_of}
        # reference route: extract the previous stage's state dicts, grow them into an l_new-shaped network
        prev_sds = [elastic.export_state_dict(e, old_mask) for e in ema]
        want_model = growth.grow_clone_ema(_fake_sd(l_new, -1.0), prev_sds[3], prev_sds[:3])
        want_ema1 = growth.grow_clone_ema(_fake_sd(l_new, -1.0), prev_sds[1], [prev_sds[1]] * 3)
        got_model = elastic.export_state_dict(model_new, new_mask) if l_new < 18 else model_new
        got_ema1 = elastic.export_state_dict(ema1_new, new_mask) if l_new < 18 else ema1_new
        assert set(got_model) == set(want_model)
        for k in want_model:
            assert torch.equal(got_model[k], want_model[k]), (l_prev, l_new, k)
            assert torch.equal(got_ema1[k], want_ema1[k]), (l_prev, l_new, k)
