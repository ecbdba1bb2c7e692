"""CPU tests of the product's host-side logic (no GPU, no kernel launches): integer bookkeeping
against the reference golden tables (bit exact), constructor/registry API, state-dict compatibility
with reference-produced state dicts, and the C ABI surface of the built library."""
import os
import re
import types

import numpy as np
import pytest
import torch

from tests._golden import load, sub

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ------------------------------------------------------------------ integer bookkeeping (bit exact)
def test_make_divisible_and_index_maps_match_reference():
    from autoprog_amd.prog import make_divisible, new_idx, get_new_layer_idx
    g = load("int_tables")
    got = np.array([[make_divisible(float(v), int(d)) for d in g["md_div"]] for v in g["md_v"]])
    assert np.array_equal(got, g["md_out"])
    for (prev, new), row, fresh in zip(g["ni_pairs"], g["ni_map"], g["ni_fresh"]):
        prev, new = int(prev), int(new)
        assert [new_idx(i, prev, new) for i in range(new)] == list(row[:new])
        assert get_new_layer_idx(prev, new) == [int(v) for v in fresh if v >= 0]


def test_active_layer_mask_matches_reference_set_sample_config():
    from autoprog_amd.prog import ActiveLayerMask
    g = load("int_tables")
    for (l, lmin, lmax, d0, d1), mask in zip(g["ss_cfg"], g["ss_mask"]):
        m = ActiveLayerMask(int(l), int(lmin), int(lmax))
        flags = [int(m.is_identity(0, i)) for i in range(d0)] + [int(m.is_identity(1, i)) for i in range(d1)]
        assert flags == [int(v) for v in mask if v >= 0], (l, lmin, lmax)
    # SURVEY.md pinned values on h12_l18 with min 9 / max 18
    assert sorted(ActiveLayerMask(9, 9, 18).skip[0]) == [1, 3] and sorted(ActiveLayerMask(9, 9, 18).skip[1]) == [1, 3, 5, 7, 9, 11, 13]
    assert sorted(ActiveLayerMask(12, 9, 18).skip[1]) == [1, 3, 5, 7, 9, 11] and not ActiveLayerMask(12, 9, 18).skip[0]
    assert sorted(ActiveLayerMask(15, 9, 18).skip[1]) == [1, 3, 5] and not any(ActiveLayerMask(18, 9, 18).skip)


def test_model_set_sample_config_flags():
    from autoprog_amd.models import create_model
    m = create_model("model_variant", variant="volo_h2_l18", num_classes=8)
    m.set_sample_config(dict(layer_num=12, min_layer_num=9, max_layer_num=18))
    assert [int(b.is_identity_layer) for b in m.network[0]] == [0, 0, 0, 0]
    assert [int(b.is_identity_layer) for b in m.network[2]] == [0, 1, 0, 1, 0, 1, 0, 1, 0, 1, 0, 1, 0, 0]


def test_rand_bbox_reproduces_reference_rng_sequence():
    from autoprog_amd.models.volo import rand_bbox
    g = load("int_tables")
    for (seed, grid), row in zip(g["bb_seed"], g["bb_out"]):
        np.random.seed(int(seed))
        lam = np.random.beta(1.0, 1.0)
        assert lam == row[0]
        assert list(rand_bbox((4, int(grid), int(grid), 8), lam, scale=2)) == [int(v) for v in row[1:]]


@pytest.mark.parametrize("tag,kw", [("script", dict(aa_scale=0.5, dp_scale=0.0, re_scale=0.0, epochs=100)), ("default", {}),
                                    ("s3", dict(num_stages=3, epochs=90, r_scale=0.6, l_scale=0.4))])
def test_progressive_schedule_matches_reference(tag, kw):
    from autoprog_amd.prog import progressive_schedule
    g = load("int_tables")
    a = types.SimpleNamespace(num_stages=4, r_scale=0.5, h_scale=1.0, l_scale=0.5, aa_scale=0.0, dp_scale=-0.5, re_scale=-0.5,
                              resize_scale=[1.0, 1.0], aa="rand-m9-mstd0.5-inc1", drop_path=0.1, reprob=0.25, scale=[0.08, 1.0], epochs=300)
    for k, v in kw.items():
        setattr(a, k, v)
    e, r, h, l, aa, dp, re_, rs = progressive_schedule(a, r_max=224, h_max=12, l_max=18)
    mags = [int(s.split("-")[1].lstrip("m")) if s else 0 for s in aa]
    for nm, val in zip(("e", "r", "h", "l", "aa"), (e, r, h, l, mags)):
        assert list(val) == [int(v) for v in g["ps_%s_%s" % (tag, nm)]], nm
    for nm, val in zip(("dp", "re", "rs"), (dp, re_, rs)):
        assert np.array_equal(np.array(val), g["ps_%s_%s" % (tag, nm)]), nm


# ------------------------------------------------------------------ constructor / registry API
def test_registry_and_parameter_counts():
    from autoprog_amd.models import create_model, list_models
    names = list_models()
    for n in ["model_variant", "volo_d1", "volo_d2", "volo_d3", "volo_d4", "volo_d5", "deit_tiny_patch16_224", "deit_base_distilled_patch16_384"]:
        assert n in names
    # timm semantics: None-valued kwargs are dropped, drop_connect_rate aliases drop_path_rate
    m = create_model("model_variant", variant="volo_h12_l18", pretrained=False, num_classes=None, drop_rate=0.0, drop_connect_rate=None,
                     drop_path_rate=0.1, drop_block_rate=None, global_pool=None, bn_tf=False, bn_momentum=None, bn_eps=None, img_size=None)
    assert sum(p.numel() for p in m.parameters()) == 26632040          # exact VOLO-D1 size
    assert len(m.state_dict()) == 260                                   # SURVEY.md row A14 (incl. BN buffers)
    assert m.num_classes == 1000 and m.no_weight_decay() == {"pos_embed", "cls_token"}
    assert abs(m.network[2][13].drop_prob - 0.1) < 1e-12 and m.network[2][0].drop_prob == pytest.approx(0.1 * 4 / 17)
    from autoprog_amd.models.volo import volo_d1, volo_d5
    assert sum(p.numel() for p in volo_d1().parameters()) == 26632040
    assert sum(p.numel() for p in volo_d5(img_size=448).parameters()) == 295907168  # exact VOLO-D5 size (448 px pos-embed)
    with pytest.raises(RuntimeError):
        create_model("no_such_model")


def test_state_dict_keys_match_reference_checkpoints():
    """state dicts produced by the REAL reference load strictly into the mirror modules"""
    from autoprog_amd.models import create_model
    d = load("volo_full")
    for tag, variant, classes in [("h2_l3", "volo_h2_l3", 16), ("h2_l6", "volo_h2_l6", 12)]:
        model = create_model("model_variant", variant=variant, num_classes=classes, img_size=64, stem_hidden_dim=16)
        sd = {k: torch.from_numpy(np.asarray(v)) for k, v in sub(d, tag + ".w").items()}
        res = model.load_state_dict(sd, strict=True)
        assert not res.missing_keys and not res.unexpected_keys


def test_deit_variants_construct():
    from autoprog_amd.models import create_model
    m = create_model("model_variant", variant="deit_h3_l4")
    assert sum(p.numel() for p in m.parameters()) == 2158504
    skip = m.set_sample_config(dict(layer_num=3, min_layer_num=2, max_layer_num=4))
    assert skip == [1]
    t = create_model("deit_tiny_distilled_patch16_224")
    assert t.pos_embed.shape == (1, 198, 192) and hasattr(t, "head_dist")


def test_models_refuse_cpu_tensors():
    from autoprog_amd.models import create_model
    m = create_model("model_variant", variant="volo_h2_l3", num_classes=8, img_size=64, stem_hidden_dim=16)
    with pytest.raises(RuntimeError):
        m(torch.zeros(1, 3, 64, 64))


# ------------------------------------------------------------------ C ABI surface
def test_library_exports_every_declared_symbol():
    import ctypes
    from autoprog_amd._lib import LIB_PATH, EXPORTED_SYMBOLS, lib
    header = open(os.path.join(ROOT, "include", "autoprog_hip.h")).read()
    declared = set(re.findall(r"\b(ap_[a-z0-9_]+)\s*\(", header))
    assert declared, "no entry points parsed from the header"
    assert declared == set(EXPORTED_SYMBOLS), (declared ^ set(EXPORTED_SYMBOLS))
    raw = ctypes.CDLL(LIB_PATH)
    for sym in declared:
        assert hasattr(raw, sym), sym
    assert lib.ap_abi_version() == 7
    assert lib.ap_error_string(-2).decode().startswith("configuration not supported")


@pytest.mark.parametrize("world", [2, 8])
def test_bench_self_launch_gloo(world):
    """`python bench.py --gpus N` without a launcher starts torch.distributed.run itself (the reference starts its ranks from
    distributed_train_prog.sh:4); --launch-check runs only the rendezvous and the bucketed gradient exchange (CPU, gloo).  N = 8 is the
    node the metric is quoted on: the launcher, the rendezvous and the reducer's bucket walk with eight ranks."""
    import json
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--launch-check"], capture_output=True, text=True,
                       timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["launch_check"] == "ok" and d["rccl_ranks"] == world and abs(d["grad_mean"] - (world + 1) / 2) < 1e-6


def test_drop_path_rates_follow_the_active_blocks():
    """set_drop_path_rate on a supernet config == the rates the reference constructor gives the extracted network"""
    from autoprog_amd.models import create_model
    from autoprog_amd.models.volo import Transformer
    m = create_model("model_variant", variant="volo_h12_l18", drop_path_rate=0.1)
    m.set_sample_config(dict(layer_num=12, min_layer_num=9, max_layer_num=18))
    m.set_drop_path_rate(0.0667)
    ext = create_model("model_variant", variant="volo_h12_l12", drop_path_rate=0.0667)
    got = [b.drop_prob for s in m.network if isinstance(s, torch.nn.Sequential) for b in s if isinstance(b, Transformer) and not b.is_identity_layer]
    want = [b.drop_prob for s in ext.network if isinstance(s, torch.nn.Sequential) for b in s if isinstance(b, Transformer)]
    assert len(got) == len(want) == 8 and all(abs(a - b) < 1e-12 for a, b in zip(got, want)), (got, want)


def test_driver_batch_splits_follow_the_reference_rule():
    """AutoProgDriver.splits_for (reference main_prog.py:567-574, 839-842: batch_splits = get_divisor(original_batch_splits,
    l r^2 / (l_max r_max^2)) -- the smallest divisor of the largest stage's split count that is > original * activation ratio)"""
    from autoprog_amd.prog.driver import AutoProgDriver
    drv = AutoProgDriver(model=None, loss_fn=None, optimizer=None, reducer=None, get_batch=None, r_list=[128, 160, 192, 224], l_list=[6, 9, 12, 12],
                         dp_list=[0.0, 0.03, 0.07, 0.1], grow_epochs=[0, 25, 50, 75], steps_per_epoch=1, original_batch_splits=4)

    def reference(number, factor):                      # main_prog.py:2057-2061, restated
        for i in range(int(number * factor) + 1, number + 1):
            if number % i == 0:
                return i
        return number
    for l, r in [(6, 128), (9, 160), (12, 192), (12, 224), (6, 224), (12, 128)]:
        assert drv.splits_for(l, r) == reference(4, (l * r * r) / (12 * 224 * 224)), (l, r)
    assert drv.splits_for(12, 224) == 4 and drv.splits_for(6, 128) == 1
    assert AutoProgDriver(None, None, None, None, None, [64], [3], [0.0], [0], 1).splits_for(3, 64) == 1      # default: no splits


def test_driver_search_after_a_stage_with_batch_splits():
    """a search that follows a training stage run at k >= 2 micro-batches per update.  `_time` used to close its passes as
    accumulating micro-batches (no exchange, `_micro` left at 1) and the first training step of the search then died in
    set_accumulate_steps.  Real GradientBucketReducer + the real driver on a CPU stand-in model: the timed passes are updates of one
    micro-batch, every batch of the search has the reference's search size (original_batch_splits, main_prog.py:807-810), and the
    reducer ends between updates."""
    import contextlib
    from autoprog_amd.dist import GradientBucketReducer
    from autoprog_amd.prog.driver import AutoProgDriver

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.fc = torch.nn.Linear(8, 4)
            self.configs = []

        def set_sample_config(self, cfg):
            self.configs.append((cfg["layer_num"], cfg["input_size"]))

        def set_drop_path_rate(self, rate):
            pass

        def forward(self, x):
            return self.fc(x)

    class Opt:
        def __init__(self, red):
            self.red, self.steps = red, 0

        def step(self):
            assert self.red.is_update_step
            self.red.take_pending_scale()
            self.steps += 1

        def grow(self, *a, **k):
            pass

        def ema_weights(self, i):
            return contextlib.nullcontext()

    torch.manual_seed(0)
    model = Net()
    red = GradientBucketReducer(list(model.parameters()), world_size=1)
    opt = Opt(red)
    asked = []

    def get_batch(r, k=1):
        asked.append((r, k))
        return torch.randn(8 // k, 8), torch.randn(8 // k, 4)
    closed = []
    real_finish = red.finish
    red.finish = lambda: (real_finish(), closed.append(red.is_update_step))[0]
    drv = AutoProgDriver(model, lambda out, t: ((out - t) ** 2).mean(), opt, red, get_batch, r_list=[64, 80, 96], l_list=[3, 4, 6],
                         dp_list=[0.0, 0.0, 0.0], grow_epochs=[0, 2, 4], steps_per_epoch=2, search_epochs=1, probe_batches=1, time_steps=1,
                         original_batch_splits=4)
    drv._transition(6, 96, 0.0)
    assert drv.batch_splits == 4
    drv._train_step(6, 96, 0.0)                                   # an update of four micro-batches
    assert asked == [(96, 4)] * 4 and closed == [False, False, False, True] and opt.steps == 1
    asked.clear(); closed.clear()
    drv._transition(3, 64, 0.0)
    r, l = drv.search(1, 2)                                       # died with "set_accumulate_steps() inside an update" before the fix
    assert (r, l) in [(rr, ll) for rr in (64, 80) for ll in (4, 6)]
    assert asked and all(k == 4 for _, k in asked), asked          # timing, probes and training steps of a search: the search's micro-batch
    n_time = 4 * 2                                                # four candidates x (1 warm-up + 1 timed) passes
    assert closed[:n_time] == [True] * n_time                     # each timed pass is a whole update (exchange included)
    train = closed[n_time:]
    assert len(train) == 2 * 4 and train == [False, False, False, True] * 2
    assert red._micro in (0, red.accumulate_steps)
    red.zero_grad()
    red.set_accumulate_steps(1)                                   # zero_grad() abandons an open update: the split count may change
    red.set_accumulate_steps(3)
    red.finish()
    with pytest.raises(RuntimeError):
        red.set_accumulate_steps(2)                               # one of three micro-batches closed: inside an update


def test_bicubic_tap_matrices_equal_torch_interpolate():
    """functional.bicubic_tap_matrix (the tap matrices ap_resample_grid multiplies the position embedding by) against
    torch.nn.functional.interpolate(scale_factor=(h0 + 0.1) / h, mode="bicubic") -- the call of VOLO.interpolate_pos_encoding
    (models/volo.py:580-596) -- down- and upsampling, square and not: float32-level agreement with the fp64 and the fp32 torch result"""
    import torch.nn.functional as F
    from autoprog_amd.functional import bicubic_tap_matrix
    g = torch.Generator().manual_seed(0)
    for (h, h0) in [(14, 8), (14, 10), (14, 12), (14, 14), (14, 20), (14, 28), (28, 14), (7, 9), (12, 5)]:
        for (w, w0) in [(14, 8), (14, 12), (14, 21), (9, 14)]:
            pos = torch.randn(1, 6, h, w, dtype=torch.float64, generator=g)
            ref = F.interpolate(pos, scale_factor=((h0 + 0.1) / h, (w0 + 0.1) / w), mode="bicubic")
            assert tuple(ref.shape[-2:]) == (h0, w0)
            wy = torch.from_numpy(bicubic_tap_matrix(h, h0, (h0 + 0.1) / h)).double()
            wx = torch.from_numpy(bicubic_tap_matrix(w, w0, (w0 + 0.1) / w)).double()
            assert wy.shape == (h0, h) and int((wy != 0).sum(1).max()) <= 4
            got = torch.einsum("oi,pj,ncij->ncop", wy, wx, pos)
            assert float((got - ref).abs().max() / ref.abs().max()) < 5e-6, (h, h0, w, w0)
