"""Pin the CPU oracle (oracle/ref_cpu.py) against golden vectors produced by the real
reference (tools/gen_golden.py).  Tolerances: integer rows exact; fp32 rows <= 1e-5 rel
(SURVEY.md section 8(c) row O5)."""
import numpy as np
import pytest
import torch

from oracle import ref_cpu as R
from tests._golden import load, sub, tensors, rel_err, max_err

TOL = 1e-5


def _params(d, prefix, dtype=torch.float64, grad=True):
    out = {}
    for k, v in sub(d, prefix + ".w").items():
        t = torch.from_numpy(v)
        if t.dtype.is_floating_point:
            t = t.to(dtype)
            if grad:
                t.requires_grad_(True)
        out[k] = t
    return out


# ------------------------------------------------------------------ integer rows (bit exact)
def test_make_divisible_table():
    g = load("int_tables")
    got = np.array([[R.make_divisible(float(v), int(d)) for d in g["md_div"]] for v in g["md_v"]])
    assert np.array_equal(got, g["md_out"])


def test_new_idx_tables():
    g = load("int_tables")
    for (prev, new), row, fresh in zip(g["ni_pairs"], g["ni_map"], g["ni_fresh"]):
        prev, new = int(prev), int(new)
        assert [R.new_idx(i, prev, new) for i in range(new)] == list(row[:new])
        assert R.get_new_layer_idx(prev, new) == [int(v) for v in fresh if v >= 0]


def test_survey_pinned_index_values():
    assert R.get_new_layer_idx(7, 14) == [1, 3, 5, 7, 9, 11, 13]
    assert R.get_new_layer_idx(2, 4) == [1, 3]
    assert R.get_new_layer_idx(8, 14) == [3, 5, 7, 9, 11, 13]
    assert R.get_new_layer_idx(11, 14) == [9, 11, 13]
    assert [R.stage_depths(l) for l in (9, 12, 15, 18)] == [[2, 7, 0, 0], [4, 8, 0, 0], [4, 11, 0, 0], [4, 14, 0, 0]]


def test_skip_masks():
    g = load("int_tables")
    for (l, lmin, lmax, d0, d1), mask in zip(g["ss_cfg"], g["ss_mask"]):
        table = R.skip_layer_table(int(l), int(lmin), int(lmax))
        flags = [int(i in table[0]) for i in range(d0)] + [int(i in table[1]) for i in range(d1)]
        assert flags == [int(v) for v in mask if v >= 0], (l, lmin, lmax)


def test_stage_depths():
    g = load("int_tables")
    for l, l0 in zip(g["depth_l"], g["depth_l0"]):
        assert R.make_divisible(int(l) * 0.23, 2) == int(l0)


def test_rand_bbox_sequences():
    g = load("int_tables")
    for (seed, grid), row in zip(g["bb_seed"], g["bb_out"]):
        rng = np.random.RandomState(int(seed))
        lam, box = R.draw_mix_box((4, int(grid), int(grid), 8), 2, 1.0, rng)
        assert lam == row[0]
        assert list(box) == [int(v) for v in row[1:]]


@pytest.mark.parametrize("tag,kw", [("script", dict(aa_scale=0.5, dp_scale=0.0, re_scale=0.0, epochs=100)), ("default", {}),
                                    ("s3", dict(num_stages=3, epochs=90, r_scale=0.6, l_scale=0.4))])
def test_progressive_schedule(tag, kw):
    g = load("int_tables")
    args = dict(num_stages=4, epochs=300, r_scale=0.5, h_scale=1.0, l_scale=0.5, aa_scale=0.0, dp_scale=-0.5, re_scale=-0.5,
                resize_scale=[1.0, 1.0], aa="rand-m9-mstd0.5-inc1", drop_path=0.1, reprob=0.25, scale=[0.08, 1.0])
    args.update(kw)
    e, r, h, l, aa, dp, re, rs = R.progressive_schedule(**args)
    mags = [int(s.split("-")[1].lstrip("m")) if s else 0 for s in aa]
    for nm, val in zip(("e", "r", "h", "l", "aa"), (e, r, h, l, mags)):
        assert list(val) == [int(v) for v in g["ps_%s_%s" % (tag, nm)]], nm
    for nm, val in zip(("dp", "re", "rs"), (dp, re, rs)):
        assert np.array_equal(np.array(val), g["ps_%s_%s" % (tag, nm)]), nm
    if tag == "script":
        assert r == [128, 160, 192, 224] and l == [9, 12, 15, 18] and e == [0, 25, 50, 75]


# ------------------------------------------------------------------ fp rows
def _check_module(d, tag, fn, wnames=None):
    p = _params(d, tag)
    x = torch.from_numpy(d[tag + ".x"]).double().requires_grad_(True)
    y = fn(x, p)
    assert rel_err(y, d[tag + ".y"]) < TOL, tag
    y.backward(torch.from_numpy(d[tag + ".dy"]).double())
    assert rel_err(x.grad, d[tag + ".dx"]) < TOL, tag
    for k, gv in sub(d, tag + ".g").items():
        assert rel_err(p[k].grad, gv) < TOL, (tag, k)


@pytest.mark.parametrize("tag", ["even8", "odd7", "rect6x10", "odd5x9", "even16"])
def test_outlook_attention(tag):
    d = load("outlook_attn")
    heads = int(d[tag + ".heads"])
    _check_module(d, tag, lambda x, p: R.outlook_attention(x, p, "", heads))


def test_blocks():
    d = load("blocks")
    H = 2
    _check_module(d, "mlp", lambda x, p: R.mlp(x, p, ""))
    _check_module(d, "attention", lambda x, p: R.attention(x.reshape(x.shape[0], -1, x.shape[-1]), p, "", H).reshape(x.shape))
    _check_module(d, "attention_n25", lambda x, p: R.attention(x.reshape(x.shape[0], -1, x.shape[-1]), p, "", H).reshape(x.shape))
    _check_module(d, "class_attention", lambda x, p: R.class_attention(x, p, "", H))
    _check_module(d, "class_block", lambda x, p: R.class_block(x, p, "", H))
    _check_module(d, "outlooker", lambda x, p: R.outlooker(x, p, "", H))
    _check_module(d, "transformer", lambda x, p: R.transformer(x, p, "", H))
    _check_module(d, "downsample", lambda x, p: R.downsample(x, p, ""))
    _check_module(d, "layernorm_1e-05", lambda x, p: R.layernorm(x, p["weight"], p["bias"], 1e-5))
    _check_module(d, "layernorm_1e-06", lambda x, p: R.layernorm(x, p["weight"], p["bias"], 1e-6))


def test_stem_train_and_eval():
    d = load("stem")
    p = _params(d, "train")
    x = torch.from_numpy(d["train.x"]).double()
    y = R.patch_embed(x, p, train=True, patch_size=8, pre="")
    assert rel_err(y.permute(0, 3, 1, 2), d["train.y"]) < TOL
    y.backward(torch.from_numpy(d["train.dy"]).double().permute(0, 2, 3, 1))
    for k, gv in sub(d, "train.g").items():
        assert rel_err(p[k].grad, gv) < 5e-5, k
    ye = R.patch_embed(x, {k: v.detach() for k, v in p.items()}, train=False, patch_size=8, pre="")
    assert rel_err(ye.permute(0, 3, 1, 2), d["eval.y"]) < TOL


@pytest.mark.parametrize("fixture", ["stem64", "stem128"])
def test_stem64_train_and_eval(fixture):
    """the BASELINE-width stem (64 channels: the shapes the HIP convolution kernels are written for) against the reference
    vectors of tests/golden/stem64.npz -- the oracle side of tests/test_gpu_blocks.py::test_hip_stem64_vs_reference_golden; round 5:
    the same for the 128-wide stem of VOLO-D4 / D5 (stem128.npz, csrc/conv128.hip)"""
    d = load(fixture)
    p = _params(d, "train")
    x = torch.from_numpy(d["train.x"]).double().requires_grad_(True)
    y = R.patch_embed(x, p, train=True, patch_size=8, pre="")
    assert rel_err(y.permute(0, 3, 1, 2), d["train.y"]) < TOL
    y.backward(torch.from_numpy(d["train.dy"]).double().permute(0, 2, 3, 1))
    assert rel_err(x.grad, d["train.dx"]) < 5e-5
    for k, gv in sub(d, "train.g").items():
        assert rel_err(p[k].grad, gv) < 5e-5, k
    ye = R.patch_embed(x.detach(), {k: v.detach() for k, v in p.items()}, train=False, patch_size=8, pre="")
    assert rel_err(ye.permute(0, 3, 1, 2), d["eval.y"]) < TOL


def test_pos_interp():
    d = load("pos_interp")
    pos = torch.from_numpy(d["pos"])
    for g in (8, 10, 12, 14, 16, 7):
        assert max_err(R.interpolate_pos_encoding(pos, g, g), d["interp_%d" % g]) < 1e-6
    pos4 = torch.from_numpy(d["pos4"])
    for g in (2, 3, 4, 6):
        assert max_err(R.interpolate_pos_encoding(pos4, g, g), d["interp4_%d" % g]) < 1e-6


@pytest.mark.parametrize("tag,variant", [("h2_l3", "volo_h2_l3"), ("h2_l6", "volo_h2_l6"), ("h2_l3_s64", "volo_h2_l3")])
def test_volo_full_train_eval(tag, variant):
    """whole networks of the reference: volo_full.npz (16-wide stem, batch 2) and -- round 5 -- volo_full64.npz: the SHIPPED 64-wide stem
    at batch 8 (tools/gen_golden.py::gen_volo_full64)"""
    d = load("volo_full64" if tag.endswith("_s64") else "volo_full")
    arch = R.variant_arch(variant)
    p = _params(d, tag)
    x = torch.from_numpy(d[tag + ".x"]).double()
    rng = np.random.RandomState(int(d[tag + ".np_seed"]))
    g1 = x.shape[-1] // 8
    lam, box = R.draw_mix_box((x.shape[0], g1, g1, arch["embed_dims"][0]), 2, 1.0, rng)
    assert lam == float(d[tag + ".lam"]) and list(box) == [int(v) for v in d[tag + ".bbox"]]
    # the fixture's state dict was captured AFTER the train forward: running stats are not inputs of train mode
    x_cls, x_aux, bb = R.volo_forward(p, x, train=True, mix=(lam, box), **arch)
    assert rel_err(x_cls, d[tag + ".x_cls"]) < TOL and rel_err(x_aux, d[tag + ".x_aux"]) < TOL
    loss = R.token_label_ce((x_cls, x_aux, bb), torch.from_numpy(d[tag + ".target"]).double(), 0.5, 1.0)
    assert abs(float(loss) - float(d[tag + ".loss"])) < 1e-6
    loss.backward()
    for k, gv in sub(d, tag + ".g").items():
        assert rel_err(p[k].grad, gv) < 1e-4, k
    ye = R.volo_forward({k: v.detach() for k, v in p.items()}, x, train=False, **arch)
    assert rel_err(ye, d[tag + ".eval_y"]) < TOL


def test_supernet_subconfigs():
    d = load("volo_full")
    arch = R.variant_arch("volo_h2_l6")
    p = _params(d, "super", grad=False)
    x = torch.from_numpy(d["super.x"]).double()
    for l in (3, 4, 5, 6):
        skip = R.skip_layer_table(l, 3, 6)
        y = R.volo_forward(p, x, train=False, skip=skip, **arch)
        assert rel_err(y, d["super.eval_y_l%d" % l]) < TOL, l


def test_losses():
    d = load("loss")
    for tag, fn, tkey, dw, cw in [("tl_box", R.token_label_ce, "t3", 0.5, 1.0), ("tl_nobox", R.token_label_ce, "t3", 0.5, 1.0),
                                  ("tl_2d", R.token_label_ce, "t2", 1.0, 1.0), ("gt_box", R.token_label_gt_ce, "t3", 0.5, 1.0),
                                  ("gt_2d", R.token_label_gt_ce, "t2", 0.5, 1.0)]:
        cls = torch.from_numpy(d["cls"]).double().requires_grad_(True)
        aux = torch.from_numpy(d["aux"]).double().requires_grad_(True)
        bb = tuple(int(v) for v in d[tag + ".bbox"])
        loss = fn((cls, aux, bb), torch.from_numpy(d[tkey]).double(), dw, cw)
        assert abs(float(loss) - float(d[tag + ".loss"])) < 1e-6, tag
        loss.backward()
        assert rel_err(cls.grad, d[tag + ".dcls"]) < TOL and rel_err(aux.grad, d[tag + ".daux"]) < TOL, tag
    x = torch.from_numpy(d["st.x"]).double().requires_grad_(True)
    loss = R.soft_target_ce(x, torch.from_numpy(d["t2"]).double())
    assert abs(float(loss) - float(d["st.loss"])) < 1e-6
    loss.backward()
    assert rel_err(x.grad, d["st.dx"]) < TOL
    x = torch.from_numpy(d["tlst.x"]).double().requires_grad_(True)
    loss = R.token_label_soft_target_ce(x, torch.from_numpy(d["tlst.t"]).double())
    assert abs(float(loss) - float(d["tlst.loss"])) < 1e-6
    loss.backward()
    assert rel_err(x.grad, d["tlst.dx"]) < TOL


def test_deit_block_matches_pinned_volo_block():
    """DeiT arithmetic is un-vendored (parity unpinned); its block equals the pinned VOLO
    Transformer math up to layout / eps, so cross-check vit_block against it."""
    d = load("blocks")
    p = _params(d, "transformer", grad=False)
    x = torch.from_numpy(d["transformer.x"]).double()
    B, H, W, C = x.shape
    y = R.vit_block(x.reshape(B, H * W, C), p, "", 2, eps=1e-5).reshape(B, H, W, C)
    assert rel_err(y, d["transformer.y"]) < TOL


def test_loss_curve_five_adamw_steps():
    """the loss-curve pin: 5 AdamW steps of the tiny model on a fixed batch (reference run)"""
    d = load("step_curve")
    arch = R.variant_arch("volo_h2_l3")
    p = {k[2:]: torch.from_numpy(v).double() for k, v in d.items() if k.startswith("w.") and v.dtype.kind == "f"}
    train = {k: v.requires_grad_(True) for k, v in p.items() if "running_" not in k}
    decay = [v for k, v in train.items() if not (v.dim() == 1 or k.endswith(".bias") or k in ("pos_embed", "cls_token"))]
    no_decay = [v for k, v in train.items() if (v.dim() == 1 or k.endswith(".bias") or k in ("pos_embed", "cls_token"))]
    opt = torch.optim.AdamW([{"params": decay, "weight_decay": float(d["wd"])}, {"params": no_decay, "weight_decay": 0.0}], lr=float(d["lr"]))
    x = torch.from_numpy(d["x"]).double()
    t = torch.from_numpy(d["target"]).double()
    rng = np.random.RandomState(int(d["np_seed"]))
    for step in range(5):
        lam, box = R.draw_mix_box((x.shape[0], 8, 8, 32), 2, 1.0, rng)
        assert list(box) == [int(v) for v in d["boxes"][step]]
        loss = R.token_label_ce(R.volo_forward(p, x, train=True, mix=(lam, box), **arch), t, 0.5, 1.0)
        assert abs(float(loss.detach()) - float(d["losses"][step])) < 2e-4, step
        opt.zero_grad()
        loss.backward()
        opt.step()


def test_loss_curve_realistic_init_ten_adamw_steps():
    """the realistic-init pin (tests/golden/step_curve_init.npz: reference volo_h4_l6 run in fp64, weights with the statistics
    of the reference's own _init_weights regenerated from a seed, 10 AdamW steps): fp64 oracle losses <= 1e-8 abs, first-step
    gradients of the sampled tensors <= 1e-6 rel (stored as fp32), gradient norms of ALL tensors <= 1e-9 rel."""
    from tests._initweights import init_state_dict
    from autoprog_amd.models import create_model
    d = load("step_curve_init")
    classes = int(d["classes"])
    arch = R.variant_arch("volo_h4_l6")
    shapes = create_model("model_variant", variant="volo_h4_l6", num_classes=classes, img_size=64, stem_hidden_dim=16).state_dict()
    sd = init_state_dict(shapes, int(d["init_seed"]))
    p = {k: v.double() for k, v in sd.items() if v.dtype.is_floating_point}
    train = {k: v.requires_grad_(True) for k, v in p.items() if "running_" not in k}
    nd = lambda k, v: v.dim() == 1 or k.endswith(".bias") or k in ("pos_embed", "cls_token")
    opt = torch.optim.AdamW([{"params": [v for k, v in train.items() if not nd(k, v)], "weight_decay": float(d["wd"])},
                             {"params": [v for k, v in train.items() if nd(k, v)], "weight_decay": 0.0}], lr=float(d["lr"]))
    x = torch.from_numpy(d["x"]).double()
    t = torch.from_numpy(d["target"]).double()
    rng = np.random.RandomState(int(d["np_seed"]))
    for step in range(10):
        lam, box = R.draw_mix_box((x.shape[0], 8, 8, 64), 2, 1.0, rng)
        assert list(box) == [int(v) for v in d["boxes"][step]]
        loss = R.token_label_ce(R.volo_forward(p, x, train=True, mix=(lam, box), **arch), t, 0.5, 1.0)
        assert abs(float(loss.detach()) - float(d["losses"][step])) < 1e-8, (step, float(loss.detach()), float(d["losses"][step]))
        opt.zero_grad()
        loss.backward()
        if step == 0:
            for k, v in d.items():
                if k.startswith("g0.") :
                    assert rel_err(train[k[3:]].grad, v) < 1e-6, k
            for name, gn in zip(d["g0_norms_names"], d["g0_norms"]):
                assert abs(float(train[str(name)].grad.norm()) - float(gn)) <= 1e-9 * float(gn) + 1e-15, name
        opt.step()


def test_vit_block_against_an_independent_statement():
    """DeiT rows D2 / D3: timm 0.4.5 is not under /root/reference, so the oracle's ViT block (oracle/ref_cpu.py vit_block) cannot be
    pinned against reference vectors.  Second, independent statement of the same pre-LN block: torch.nn.MultiheadAttention (packed
    in_proj = qkv, out_proj = proj -- torch's own attention code path, none of the oracle's) + nn.LayerNorm + nn.Linear / exact-erf
    nn.GELU, in fp64, wired x + attn(LN1 x), x + mlp(LN2 x) as timm's Block.forward.  Outputs and all gradients agree to 1e-10."""
    torch.manual_seed(0)
    B, N, C, heads, hidden = 3, 17, 48, 3, 192
    mha = torch.nn.MultiheadAttention(C, heads, bias=True, batch_first=True).double()
    ln1, ln2 = torch.nn.LayerNorm(C, eps=1e-6).double(), torch.nn.LayerNorm(C, eps=1e-6).double()
    fc1, fc2, act = torch.nn.Linear(C, hidden).double(), torch.nn.Linear(hidden, C).double(), torch.nn.GELU()
    for m in (mha, ln1, ln2, fc1, fc2):
        for q in m.parameters():
            torch.nn.init.normal_(q, std=0.3)
    x = torch.randn(B, N, C, dtype=torch.float64, requires_grad=True)
    h = ln1(x)
    y1 = x + mha(h, h, h, need_weights=False)[0]
    y = y1 + fc2(act(fc1(ln2(y1))))
    dy = torch.randn_like(y)
    y.backward(dy)
    p = {"b.norm1.weight": ln1.weight, "b.norm1.bias": ln1.bias, "b.norm2.weight": ln2.weight, "b.norm2.bias": ln2.bias,
         "b.attn.qkv.weight": mha.in_proj_weight, "b.attn.qkv.bias": mha.in_proj_bias,
         "b.attn.proj.weight": mha.out_proj.weight, "b.attn.proj.bias": mha.out_proj.bias,
         "b.mlp.fc1.weight": fc1.weight, "b.mlp.fc1.bias": fc1.bias, "b.mlp.fc2.weight": fc2.weight, "b.mlp.fc2.bias": fc2.bias}
    po = {k: v.detach().clone().requires_grad_(True) for k, v in p.items()}
    xo = x.detach().clone().requires_grad_(True)
    yo = R.vit_block(xo, po, "b.", heads, eps=1e-6)
    yo.backward(dy)
    assert rel_err(yo, y.detach()) < 1e-10
    assert rel_err(xo.grad, x.grad) < 1e-10
    for k in p:
        assert rel_err(po[k].grad, p[k].grad) < 1e-10, k


def _late_state_reference(d):
    names = [str(n) for n in d["grad_names"]]
    return names, {k: torch.from_numpy(d["g16." + k].astype(np.float64)) * float(d["gs." + k]) for k in names}


def test_late_state_outputs_loss_and_gradients():
    """the oracle on a TRAINED state (tests/golden/late_state.npz: the reference's volo_h4_l6 after 300 fp64 AdamW steps, evaluated by
    the reference in fp64): logits <= 1e-9 rel, loss <= 1e-9, every parameter gradient <= 2e-3 rel (the fixture stores gradients as fp16
    mantissas with a per-tensor scale) and all gradients as one vector <= 5e-4."""
    d = load("late_state")
    arch = R.variant_arch("volo_h4_l6")
    p = {k[2:]: torch.from_numpy(np.asarray(v)).double() for k, v in d.items() if k.startswith("w.") and np.asarray(v).dtype.kind == "f"}
    for k, v in p.items():
        if "running_" not in k:
            v.requires_grad_(True)
    x = torch.from_numpy(d["x"]).double()
    t = torch.from_numpy(d["target"]).double()
    lam, box = R.draw_mix_box((x.shape[0], 8, 8, 64), 2, 1.0, np.random.RandomState(int(d["np_seed"])))
    assert list(box) == [int(v) for v in d["box"]]
    out = R.volo_forward(p, x, train=True, mix=(lam, box), **arch)
    assert rel_err(out[0], d["y_cls"]) < 1e-9 and rel_err(out[1], d["y_aux"]) < 1e-9
    loss = R.token_label_ce(out, t, 0.5, 1.0)
    assert abs(float(loss.detach()) - float(d["loss"])) < 1e-9
    loss.backward()
    names, ref = _late_state_reference(d)
    for k in names:
        assert rel_err(p[k].grad, ref[k]) < 2e-3, k
    va = torch.cat([p[k].grad.flatten() for k in names]); vb = torch.cat([ref[k].flatten() for k in names])
    assert float((va - vb).norm() / vb.norm()) < 5e-4


def test_vit_and_distilled_deit_against_the_transformers_library():
    """Rows D2 / D3: the reference's DeiT (models/deit.py:20-59) sits on timm 0.4.5's VisionTransformer, which is not under /root/reference,
    so no reference-held vector constrains the oracle's `vit_forward` (parity unpinned -- stays so).  The closest independent statement
    in this image: the `transformers` library's DeiT (the port of facebook/deit, the model family models/deit.py registers) -- whole
    network: patch embedding, class + distillation tokens, position embedding, pre-LN blocks with separate q / k / v projections, final
    LayerNorm, the two heads and their eval-mode average.  Same weights (the packed qkv of the timm layout split into the three
    projections), fp64, 2 x 32 px images: class / distillation logits, their eval average and every parameter gradient agree to 1e-9."""
    transformers = pytest.importorskip("transformers")
    from transformers import DeiTConfig, DeiTForImageClassificationWithTeacher
    torch.manual_seed(0)
    C, depth, heads, hidden, classes, img, patch = 48, 3, 3, 192, 10, 32, 16
    cfg = DeiTConfig(hidden_size=C, num_hidden_layers=depth, num_attention_heads=heads, intermediate_size=hidden, image_size=img, patch_size=patch,
                     num_labels=classes, hidden_act="gelu", layer_norm_eps=1e-6, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, qkv_bias=True)
    hf = DeiTForImageClassificationWithTeacher(cfg).double().eval()
    with torch.no_grad():
        for q in hf.parameters():
            q.copy_(torch.randn(q.shape, dtype=torch.float64) * 0.2)
    sd = hf.state_dict()
    keys = set(sd.keys())
    # timm-layout parameter dict of the oracle from the library's tensors
    def pick(*names):
        for n in names:
            if n in sd:
                keys.discard(n)
                return sd[n].detach().clone()
        raise KeyError(names)
    p = {"cls_token": pick("deit.embeddings.cls_token"), "dist_token": pick("deit.embeddings.distillation_token"),
         "pos_embed": pick("deit.embeddings.position_embeddings"),
         "patch_embed.proj.weight": pick("deit.embeddings.patch_embeddings.projection.weight"),
         "patch_embed.proj.bias": pick("deit.embeddings.patch_embeddings.projection.bias"),
         "norm.weight": pick("deit.layernorm.weight"), "norm.bias": pick("deit.layernorm.bias"),
         "head.weight": pick("cls_classifier.weight"), "head.bias": pick("cls_classifier.bias"),
         "head_dist.weight": pick("distillation_classifier.weight"), "head_dist.bias": pick("distillation_classifier.bias")}
    for i in range(depth):
        L, A = "deit.layers.%d." % i, ("deit.layers.%d.attention." % i)
        alt = "deit.encoder.layer.%d." % i                       # older releases of the library
        q = pick(A + "q_proj.weight", alt + "attention.attention.query.weight"); k = pick(A + "k_proj.weight", alt + "attention.attention.key.weight")
        v = pick(A + "v_proj.weight", alt + "attention.attention.value.weight")
        qb = pick(A + "q_proj.bias", alt + "attention.attention.query.bias"); kb = pick(A + "k_proj.bias", alt + "attention.attention.key.bias")
        vb = pick(A + "v_proj.bias", alt + "attention.attention.value.bias")
        b = "blocks.%d." % i
        p[b + "attn.qkv.weight"], p[b + "attn.qkv.bias"] = torch.cat([q, k, v], 0), torch.cat([qb, kb, vb], 0)
        p[b + "attn.proj.weight"], p[b + "attn.proj.bias"] = pick(A + "o_proj.weight", alt + "attention.output.dense.weight"), pick(A + "o_proj.bias", alt + "attention.output.dense.bias")
        p[b + "norm1.weight"], p[b + "norm1.bias"] = pick(L + "layernorm_before.weight", alt + "layernorm_before.weight"), pick(L + "layernorm_before.bias", alt + "layernorm_before.bias")
        p[b + "norm2.weight"], p[b + "norm2.bias"] = pick(L + "layernorm_after.weight", alt + "layernorm_after.weight"), pick(L + "layernorm_after.bias", alt + "layernorm_after.bias")
        p[b + "mlp.fc1.weight"], p[b + "mlp.fc1.bias"] = pick(L + "mlp.fc1.weight", alt + "intermediate.dense.weight"), pick(L + "mlp.fc1.bias", alt + "intermediate.dense.bias")
        p[b + "mlp.fc2.weight"], p[b + "mlp.fc2.bias"] = pick(L + "mlp.fc2.weight", alt + "output.dense.weight"), pick(L + "mlp.fc2.bias", alt + "output.dense.bias")
    assert not keys, ("library tensors the mapping did not consume", sorted(keys))
    for t in p.values():
        t.requires_grad_(True)
    x = torch.randn(2, 3, img, img, dtype=torch.float64)
    out = hf(pixel_values=x)
    y, yd = R.vit_forward(p, x, depth=depth, heads=heads, patch=patch, distilled=True, train=True)
    assert rel_err(y, out.cls_logits) < 1e-9 and rel_err(yd, out.distillation_logits) < 1e-9
    assert rel_err(R.vit_forward(p, x, depth=depth, heads=heads, patch=patch, distilled=True, train=False), out.logits) < 1e-9
    w = torch.randn(2, classes, dtype=torch.float64)
    ((out.cls_logits * w).sum() + (out.distillation_logits * w.flip(0)).sum()).backward()
    ((y * w).sum() + (yd * w.flip(0)).sum()).backward()
    hg = {n: q.grad for n, q in hf.named_parameters()}
    assert rel_err(p["head.weight"].grad, hg["cls_classifier.weight"]) < 1e-9
    assert rel_err(p["pos_embed"].grad, hg["deit.embeddings.position_embeddings"]) < 1e-9
    assert rel_err(p["patch_embed.proj.weight"].grad, hg["deit.embeddings.patch_embeddings.projection.weight"]) < 1e-9
    qn = next(n for n in hg if n.endswith(("layers.1.attention.q_proj.weight", "layer.1.attention.attention.query.weight")))
    assert rel_err(p["blocks.1.attn.qkv.weight"].grad[:C], hg[qn]) < 1e-9
    f2 = next(n for n in hg if n.endswith(("layers.2.mlp.fc2.weight", "layer.2.output.dense.weight")))
    assert rel_err(p["blocks.2.mlp.fc2.weight"].grad, hg[f2]) < 1e-9


def test_bf16_points_functions_are_the_pinned_functions_plus_rounding(monkeypatch):
    """oracle/ref_cpu.py *_bf16_points (the rounding-matched statements the GPU tests hold the kernels to) with their rounding switched
    off (BF16_POINTS_ROUND = False) must BE the functions the golden vectors pin: outputs and every gradient equal to 1e-10 in fp64 --
    the custom autograd nodes (attention with dS, the outlook core by unfold / fold, the stored gelu') against plain autograd of the
    closed forms -- for each block type, the whole VOLO training forward and the ViT."""
    from autoprog_amd.models import create_model
    monkeypatch.setattr(R, "BF16_POINTS_ROUND", False)
    torch.manual_seed(0)

    def params_of(model):
        return {k: (v.detach().double() + 0.05 * torch.randn(v.shape, dtype=torch.float64)).requires_grad_(True) if v.dtype.is_floating_point
                else v.detach().clone() for k, v in model.state_dict().items()}

    def compare(run_plain, run_points, p):
        outs = []
        for run in (run_plain, run_points):
            for v in p.values():
                if v.dtype.is_floating_point:
                    v.grad = None
            y = run()
            ys = y if isinstance(y, (tuple, list)) else (y,)
            sum((t * torch.linspace(0.5, 1.5, t.numel(), dtype=torch.float64).reshape(t.shape)).sum() for t in ys if torch.is_tensor(t)).backward()
            outs.append(([t.detach().clone() for t in ys if torch.is_tensor(t)],
                         {k: v.grad.clone() for k, v in p.items() if v.dtype.is_floating_point and v.grad is not None}))
        (ya, ga), (yb, gb) = outs
        for a, b in zip(ya, yb):
            assert rel_err(b, a) < 1e-10
        assert set(ga) == set(gb)
        for k in ga:
            assert rel_err(gb[k], ga[k]) < 1e-9, k

    m = create_model("model_variant", variant="volo_h4_l6", num_classes=24, img_size=64)
    p = params_of(m)
    arch = R.variant_arch("volo_h4_l6")
    img = torch.randn(2, 3, 64, 64, dtype=torch.float64)
    mix = (0.7, (1, 0, 3, 2))
    compare(lambda: R.volo_forward(p, img, train=True, mix=mix, **arch)[:2],
            lambda: R.volo_forward(p, img, train=True, mix=mix, bf16_points=True, **arch)[:2], p)
    x = torch.randn(2, 7, 9, 64, dtype=torch.float64)                 # odd grid: clipped pooling windows, ragged outlook windows
    compare(lambda: R.outlooker(x, p, "network.0.0.", 2), lambda: R.outlooker_bf16_points(x, p, "network.0.0.", 2), p)
    t = torch.randn(2, 3, 5, 128, dtype=torch.float64)
    compare(lambda: R.transformer(t, p, "network.2.0.", 4), lambda: R.transformer_bf16_points(t.reshape(2, 15, 128), p, "network.2.0.", 4).reshape(2, 3, 5, 128), p)
    c = torch.randn(2, 16, 128, dtype=torch.float64)
    compare(lambda: R.class_block(c, p, "post_network.0.", 4)[:, :1], lambda: R.class_block_bf16_points(c, p, "post_network.0.", 4)[:, :1], p)
    d = create_model("model_variant", variant="deit_h3_l2", num_classes=16)
    pd = params_of(d)
    im = torch.randn(2, 3, 224, 224, dtype=torch.float64)
    compare(lambda: R.vit_forward(pd, im, depth=2, heads=3), lambda: R.vit_forward(pd, im, depth=2, heads=3, bf16_points=True), pd)
