"""Model-level parity on the GPU: the HIP-backed VOLO / loss modules against (a) golden vectors of
the real reference (tests/golden/volo_full.npz, step_curve.npz) and (b) the CPU oracle on the same
seeded inputs.  bf16 path tolerances, stated per check: full-network outputs <= 3e-2 rel-L2 per
tensor (bf16 storage rounding, ~0.4 % per op, accumulated over the ~40 ops of these stress-test
models whose random weights have O(1) scale; single kernels are held to 1e-2 in
test_gpu_kernels.py), parameter gradients <= 6e-2 rel-L2 (0.15 for the MIOpen bf16 conv stem that
sits below every block), loss values <= 5e-3 RELATIVE on the single-step fixtures and <= 1e-2
RELATIVE per step on the 5-step AdamW curve (losses of 5-7 on these O(1)-weight stress fixtures;
fp32 atomics in the weight-gradient kernels make the curve vary by a few 1e-3 run to run)."""
import numpy as np
import pytest
import torch

from oracle import ref_cpu as R
from tests._golden import load, sub

pytestmark = pytest.mark.gpu


def rel(a, b):
    a = torch.as_tensor(a).detach().double().cpu()
    b = torch.as_tensor(b).detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def build(variant, classes, img=64, dpr=0.0, stem=16):
    from autoprog_amd.models import create_model
    return create_model("model_variant", variant=variant, num_classes=classes, img_size=img, drop_path_rate=dpr, stem_hidden_dim=stem)


def forbid_torch_convolutions(monkeypatch):
    """any torch convolution on a GPU tensor fails the test: the network under test must run its stem on csrc/conv7.hip / conv.hip and
    the patch-addressed GEMMs (as test_hip_stem64_vs_reference_golden does for the stem alone)"""
    import torch.nn.functional as F
    real = F.conv2d

    def guarded(inp, *a, **k):
        if inp.is_cuda:
            raise AssertionError("torch conv2d called on the GPU: this network's stem must run on the HIP convolution kernels")
        return real(inp, *a, **k)
    monkeypatch.setattr(F, "conv2d", guarded)
    real_call = torch.nn.Conv2d.forward

    def guarded_module(self, inp):
        if inp.is_cuda:
            raise AssertionError("nn.Conv2d.forward called on the GPU: this network's stem must run on the HIP convolution kernels")
        return real_call(self, inp)
    monkeypatch.setattr(torch.nn.Conv2d, "forward", guarded_module)


def load_sd(model, d, prefix):
    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in sub(d, prefix + ".w").items()}
    missing, unexpected = model.load_state_dict(sd, strict=True), None
    return model


@pytest.mark.parametrize("tag,variant,classes", [("h2_l3", "volo_h2_l3", 16), ("h2_l6", "volo_h2_l6", 12)])
def test_volo_train_eval_vs_reference_golden(tag, variant, classes):
    from autoprog_amd.loss import TokenLabelCrossEntropy
    d = load("volo_full")
    model = load_sd(build(variant, classes), d, tag).cuda().train()
    x = torch.from_numpy(d[tag + ".x"]).cuda()
    target = torch.from_numpy(d[tag + ".target"]).cuda()
    np.random.seed(int(d[tag + ".np_seed"]))
    x_cls, x_aux, bb = model(x)
    assert list(bb) == [int(v) for v in d[tag + ".bbox"]]                 # mix-token bookkeeping: exact
    assert rel(x_cls, d[tag + ".x_cls"]) < 3e-2, rel(x_cls, d[tag + ".x_cls"])
    assert rel(x_aux, d[tag + ".x_aux"]) < 3e-2, rel(x_aux, d[tag + ".x_aux"])
    loss = TokenLabelCrossEntropy(dense_weight=0.5, cls_weight=1.0, classes=classes)((x_cls, x_aux, bb), target)
    assert abs(float(loss.detach()) - float(d[tag + ".loss"])) < 5e-3 * float(d[tag + ".loss"]), (float(loss.detach()), float(d[tag + ".loss"]))
    loss.backward()
    # Gradients.  These fixtures use O(1)-scale random weights, so parameter gradients that are sums with
    # heavy cancellation (LayerNorm/bias gradients of the first blocks) amplify bf16 rounding; every block
    # is held to 2.5e-2 per tensor in test_gpu_blocks.py, here the whole network is held to a global bound
    # plus a loose per-tensor sanity bound.
    num = den = 0.0
    worst = {}
    for name, p in model.named_parameters():
        g = torch.from_numpy(d[tag + ".g." + name]).double()
        diff = p.grad.detach().double().cpu() - g
        num += float(diff.pow(2).sum())
        den += float(g.pow(2).sum())
        if float(g.norm()) > 1e-6:
            worst[name] = float(diff.norm() / g.norm())
    print("volo_full %s: global gradient rel-L2 %.4f; worst tensors %s" % (tag, (num / den) ** 0.5, sorted(((round(v, 3), k) for k, v in worst.items()), reverse=True)[:6]))
    # measured: global 0.058 (h2_l3) / 0.018 (h2_l6), worst tensors 0.187 / 0.085 (first block's norm1 and the 16-wide MIOpen stem);
    # the bounds are 1.5x the measurement of each fixture
    g_bound, t_bound = {"h2_l3": (0.087, 0.28), "h2_l6": (0.027, 0.13)}.get(tag, (0.087, 0.28))
    assert (num / den) ** 0.5 < g_bound, (num / den) ** 0.5
    bad = {k: v for k, v in worst.items() if v > t_bound}
    assert not bad, sorted(bad.items(), key=lambda kv: -kv[1])[:10]
    load_sd(model, d, tag)                  # the train forward above updated the BN running stats once more
    model.eval()
    with torch.no_grad():
        y = model(x)
    assert rel(y, d[tag + ".eval_y"]) < 3e-2


def test_volo_on_the_shipped_stem_vs_reference_golden(monkeypatch):
    """VERDICT r4, missing 3: a reference-generated WHOLE network on the stem the BASELINE configs ship (stem_hidden_dim = 64:
    csrc/conv7.hip, csrc/conv.hip with BatchNorm in the staging, the patch-addressed projection) -- tests/golden/volo_full64.npz,
    tools/gen_golden.py::gen_volo_full64: the reference's volo_h2_l3 at 64 px, batch 8 (8 x 32 x 32 samples per BatchNorm channel),
    stress weights as volo_full.  No torch convolution may run.  Train outputs, loss, every parameter gradient, eval output."""
    from autoprog_amd.loss import TokenLabelCrossEntropy
    forbid_torch_convolutions(monkeypatch)
    tag, classes = "h2_l3_s64", 16
    d = load("volo_full64")
    model = load_sd(build("volo_h2_l3", classes, stem=64), d, tag).cuda().train()
    x = torch.from_numpy(d[tag + ".x"]).cuda()
    target = torch.from_numpy(d[tag + ".target"]).cuda()
    np.random.seed(int(d[tag + ".np_seed"]))
    x_cls, x_aux, bb = model(x)
    assert list(bb) == [int(v) for v in d[tag + ".bbox"]]
    e_cls, e_aux = rel(x_cls, d[tag + ".x_cls"]), rel(x_aux, d[tag + ".x_aux"])
    loss = TokenLabelCrossEntropy(dense_weight=0.5, cls_weight=1.0, classes=classes)((x_cls, x_aux, bb), target)
    loss.backward()
    num = den = 0.0
    worst = {}
    for name, p in model.named_parameters():
        g = torch.from_numpy(d[tag + ".g." + name]).double()
        diff = p.grad.detach().double().cpu() - g
        num += float(diff.pow(2).sum())
        den += float(g.pow(2).sum())
        if float(g.norm()) > 1e-6:
            worst[name] = float(diff.norm() / g.norm())
    glob = (num / den) ** 0.5
    print("volo_full64: outputs %.4f / %.4f, loss %.5f (reference %.5f), global gradient rel-L2 %.4f; worst tensors %s"
          % (e_cls, e_aux, float(loss), float(d[tag + ".loss"]), glob, sorted(((round(v, 3), k) for k, v in worst.items()), reverse=True)[:8]))
    assert e_cls < 3e-2 and e_aux < 3e-2, (e_cls, e_aux)
    assert abs(float(loss.detach()) - float(d[tag + ".loss"])) < 5e-3 * float(d[tag + ".loss"])
    assert glob < FULL64_GLOBAL_TOL, glob
    bad = {k: v for k, v in worst.items() if v > FULL64_TENSOR_TOL}
    assert not bad, sorted(bad.items(), key=lambda kv: -kv[1])[:10]
    load_sd(model, d, tag)                  # the train forward above updated the BN running stats once more
    model.eval()
    with torch.no_grad():
        y = model(x)
    assert rel(y, d[tag + ".eval_y"]) < 3e-2


# bounds of the 64-wide-stem network fixture = 1.5x the first GPU run of the test (round 5): outputs 1.3e-2 / 1.0e-2, loss 7.02132 against
# 7.02070, all gradients as one vector 0.0296, worst tensors 0.133 / 0.118 (the first two BatchNorm biases), 0.110 / 0.101 / 0.094 (stem
# convolution weights and the first BatchNorm weight), everything behind the stem <= 0.053.  (volo_full's h2_l3 on the 16-wide MIOpen stem
# at batch 2: 0.058 as one vector, worst 0.187.)
FULL64_GLOBAL_TOL, FULL64_TENSOR_TOL = 0.045, 0.2


def test_supernet_subconfigs_vs_reference_golden():
    d = load("volo_full")
    model = load_sd(build("volo_h2_l6", 10), d, "super").cuda().eval()
    x = torch.from_numpy(d["super.x"]).cuda()
    for l in (3, 4, 5, 6):
        model.set_sample_config(dict(layer_num=l, min_layer_num=3, max_layer_num=6))
        with torch.no_grad():
            y = model(x)
        assert rel(y, d["super.eval_y_l%d" % l]) < 3e-2, l


def test_loss_curve_vs_reference_golden():
    from autoprog_amd.loss import TokenLabelCrossEntropy
    d = load("step_curve")
    model = build("volo_h2_l3", 16)
    model.load_state_dict({k[2:]: torch.from_numpy(np.asarray(v)) for k, v in d.items() if k.startswith("w.")})
    model = model.cuda().train()
    x = torch.from_numpy(d["x"]).cuda()
    target = torch.from_numpy(d["target"]).cuda()
    decay, no_decay = [], []
    for n, p in model.named_parameters():
        (no_decay if (p.dim() == 1 or n.endswith(".bias") or n in ("pos_embed", "cls_token")) else decay).append(p)
    opt = torch.optim.AdamW([{"params": decay, "weight_decay": float(d["wd"])}, {"params": no_decay, "weight_decay": 0.0}], lr=float(d["lr"]))
    loss_fn = TokenLabelCrossEntropy(dense_weight=0.5, cls_weight=1.0, classes=16)
    np.random.seed(int(d["np_seed"]))
    losses = []
    for step in range(5):
        out = model(x)
        assert [int(v) for v in out[2]] == [int(v) for v in d["boxes"][step]]
        loss = loss_fn(out, target)
        opt.zero_grad()
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    diff = np.abs(np.array(losses) - d["losses"]) / d["losses"]
    print("loss curve hip:", losses, "ref:", d["losses"].tolist(), "reldiff:", diff.tolist())
    assert diff.max() < 4e-3, (losses, d["losses"].tolist())          # measured 1.0e-3 - 2.2e-3 relative (round 2 and 3 runs)


def test_d1_shapes_droppath_and_oracle_agreement():
    """D1-width blocks (C=192/384, 6/12 heads) at small batch/resolution against the oracle with
    injected DropPath masks and a fixed mix-token box."""
    from autoprog_amd.models import create_model
    torch.manual_seed(0)
    model = create_model("model_variant", variant="volo_h12_l9", num_classes=1000, img_size=224, drop_path_rate=0.2).cuda().train()
    B, r = 4, 64
    x = torch.randn(B, 3, r, r, device="cuda")
    masks = {}
    layers = model.layers
    total = sum(layers)
    rng = np.random.RandomState(5)
    for i, blk in enumerate(model.network[2]):
        rate = blk.drop_prob
        m1 = torch.from_numpy((rng.rand(B) < (1 - rate)).astype(np.float32))
        m2 = torch.from_numpy((rng.rand(B) < (1 - rate)).astype(np.float32))
        if rate > 0:
            masks[(1, i)] = (m1, m2)
            model.drop_path_rng.queue += [m1, m2]
    np.random.seed(3)
    x_cls, x_aux, bb = model(x)
    p = {k: v.detach().double().cpu() for k, v in model.state_dict().items()}
    np_rng = np.random.RandomState(3)
    lam, box = R.draw_mix_box((B, r // 8, r // 8, 192), 2, 1.0, np_rng)
    assert tuple(bb) == tuple(box)
    arch = R.variant_arch("volo_h12_l9")
    ref_cls, ref_aux, _ = R.volo_forward(p, x.double().cpu(), train=True, mix=(lam, box), dp_masks=masks, drop_path_rate=0.2, **arch)
    assert rel(x_cls, ref_cls) < 3e-2, rel(x_cls, ref_cls)
    assert rel(x_aux, ref_aux) < 3e-2, rel(x_aux, ref_aux)
    # backward with the realistic (trunc-normal 0.02) initialisation: per-tensor gradient parity
    from autoprog_amd.loss import TokenLabelCrossEntropy
    g = torch.Generator().manual_seed(1)
    target = torch.softmax(torch.randn(B, 1000, 2 + (r // 16) ** 2, generator=g) * 3, dim=1).cuda()
    loss = TokenLabelCrossEntropy(dense_weight=0.5, cls_weight=1.0)((x_cls, x_aux, bb), target)
    loss.backward()
    for v in p.values():
        if v.dtype.is_floating_point:
            v.requires_grad_(True)
    ref_out = R.volo_forward(p, x.double().cpu(), train=True, mix=(lam, box), dp_masks=masks, drop_path_rate=0.2, **arch)
    ref_loss = R.token_label_ce(ref_out, target.double().cpu(), 0.5, 1.0)
    ref_loss.backward()
    assert abs(float(loss.detach()) - float(ref_loss.detach())) < 2e-3 * float(ref_loss.detach())
    errs = {n: rel(q.grad, p[n].grad) for n, q in model.named_parameters() if float(p[n].grad.norm()) > 1e-9}
    print("D1-width grad errors: max %.4f  (%s)" % (max(errs.values()), max(errs, key=errs.get)))
    bad = {k: v for k, v in errs.items() if v > (0.12 if k.startswith("patch_embed.") else 6e-2)}
    assert not bad, sorted(bad.items(), key=lambda kv: -kv[1])[:10]


def test_whole_network_kernels_vs_fp64_with_the_same_rounding_points():
    """The WHOLE training forward / loss / backward of a VOLO (volo_h4_l6: HIP conv stem, 2 outlooker blocks, downsample, position
    embedding, 4 transformer blocks, 2 class blocks, both heads, mix-token, token-label loss) against the oracle's network with the MI355X
    pipeline's bf16 rounding points (oracle/ref_cpu.py volo_forward(bf16_points=True): the composition of the per-block functions the
    tests of test_gpu_blocks.py pin one by one).  The golden / oracle tests above bound what the bf16 RECIPE costs against fp32 / fp64
    (3e-2 outputs, 6e-2 - 0.28 gradients); this one bounds what the KERNELS add on top of that recipe through the full depth.  Fed from
    the oracle's tensor, every stage of this network reproduces the oracle's next tensor EXACTLY in the forward (tools/dbg_points.py: 0.0
    for both outlookers, the downsample, the position embedding, three of four transformer blocks, the class blocks, the final norm and
    both heads; 9e-5 for one transformer block) except the conv stem (7.9e-4: fp32 BatchNorm statistics, roundings that fall the other
    way); chained, that 8e-4 grows to 7.7e-3 / 5.6e-3 at the two outputs -- ten normalised blocks amplify any perturbation of their
    input about tenfold, the oracle's own included.  Measured on MI355X: outputs 7.7e-3 / 5.6e-3, loss 9e-5 relative, parameter
    gradients median 6.5e-3, max 1.4e-2 (a stem convolution).  Bounds (2x): 1.5e-2, 3e-4, WHOLE_NET_KERNEL_TOL = 3e-2 per tensor."""
    from autoprog_amd.models import create_model
    from autoprog_amd.loss import TokenLabelCrossEntropy
    torch.manual_seed(5)
    classes, B, r = 40, 4, 64
    model = create_model("model_variant", variant="volo_h4_l6", num_classes=classes, img_size=r).cuda().train()
    x = torch.randn(B, 3, r, r, device="cuda")
    g = torch.Generator().manual_seed(6)
    target = torch.softmax(torch.randn(B, classes, 2 + (r // 16) ** 2, generator=g) * 3, dim=1).cuda()
    p = {k: v.detach().double().cpu().clone() for k, v in model.state_dict().items()}
    np.random.seed(8)
    x_cls, x_aux, bb = model(x)
    loss = TokenLabelCrossEntropy(dense_weight=0.5, cls_weight=1.0, classes=classes)((x_cls, x_aux, bb), target)
    loss.backward()
    for v in p.values():
        if v.dtype.is_floating_point:
            v.requires_grad_(True)
    rng = np.random.RandomState(8)
    lam, box = R.draw_mix_box((B, r // 8, r // 8, 64), 2, 1.0, rng)
    assert tuple(bb) == tuple(box)
    arch = R.variant_arch("volo_h4_l6")
    ref = R.volo_forward(p, x.double().cpu(), train=True, mix=(lam, box), bf16_points=True, **arch)
    ref_loss = R.token_label_ce(ref, target.double().cpu(), 0.5, 1.0)
    ref_loss.backward()
    e_out = (rel(x_cls, ref[0]), rel(x_aux, ref[1]))
    e_loss = abs(float(loss.detach()) - float(ref_loss.detach())) / float(ref_loss.detach())
    errs = {n: rel(q.grad, p[n].grad) for n, q in model.named_parameters() if float(p[n].grad.norm()) > 1e-12}
    print("whole network vs rounding-matched fp64: outputs %.2e / %.2e, loss rel %.2e, gradients: median %.2e max %.2e (%s)"
          % (e_out[0], e_out[1], e_loss, float(np.median(list(errs.values()))), max(errs.values()), max(errs, key=errs.get)),
          sorted(((float("%.2e" % v), k) for k, v in errs.items()), reverse=True)[:6])
    assert max(e_out) < 1.5e-2, e_out
    assert e_loss < 3e-4, e_loss
    bad = {k: v for k, v in errs.items() if v > WHOLE_NET_KERNEL_TOL}
    assert not bad, sorted(bad.items(), key=lambda kv: -kv[1])[:10]


WHOLE_NET_KERNEL_TOL = 3e-2


def _d5_shapes_vs_oracle(out_tol, loss_tol, grad_tol, stem_tol, tag):
    from autoprog_amd.models.volo import VOLO
    from autoprog_amd.loss import TokenLabelCrossEntropy
    torch.manual_seed(0)
    arch = dict(layers=[1, 2, 0, 0], embed_dims=[384, 768, 768, 768], num_heads=[12, 16, 16, 16])
    model = VOLO(arch["layers"], img_size=448, num_classes=96, embed_dims=arch["embed_dims"], num_heads=arch["num_heads"],
                 mlp_ratios=[4, 4, 4, 4], downsamples=[True, False, False, False], outlook_attention=[True, False, False, False],
                 post_layers=["ca", "ca"], stem_hidden_dim=128).cuda().train()
    B, r = 1, 448
    x = torch.randn(B, 3, r, r, device="cuda")
    np.random.seed(4)
    x_cls, x_aux, bb = model(x)
    assert x_aux.shape == (B, 784, 96)
    g = torch.Generator().manual_seed(1)
    target = torch.softmax(torch.randn(B, 96, 2 + 784, generator=g) * 3, dim=1).cuda()
    loss = TokenLabelCrossEntropy(dense_weight=0.5, cls_weight=1.0, classes=96)((x_cls, x_aux, bb), target)
    loss.backward()
    p = {k: v.detach().double().cpu() for k, v in model.state_dict().items()}
    for v in p.values():
        if v.dtype.is_floating_point:
            v.requires_grad_(True)
    rng = np.random.RandomState(4)
    lam, box = R.draw_mix_box((B, r // 8, r // 8, 384), 2, 1.0, rng)
    assert tuple(bb) == tuple(box)
    torch.set_num_threads(max(1, min(32, torch.get_num_threads())))
    ref = R.volo_forward(p, x.double().cpu(), train=True, mix=(lam, box), **arch)
    ref_loss = R.token_label_ce(ref, target.double().cpu(), 0.5, 1.0)
    ref_loss.backward()
    e_out = (rel(x_cls, ref[0]), rel(x_aux, ref[1]))
    e_loss = abs(float(loss.detach()) - float(ref_loss.detach())) / float(ref_loss.detach())
    errs = {n: rel(q.grad, p[n].grad) for n, q in model.named_parameters() if float(p[n].grad.norm()) > 1e-9}
    print("D5-shape %s: outputs %.4f / %.4f, loss rel %.5f, grad errors: median %.4f max %.4f (%s)"
          % (tag, e_out[0], e_out[1], e_loss, float(np.median(list(errs.values()))), max(errs.values()), max(errs, key=errs.get)))
    assert max(e_out) < out_tol, e_out
    assert e_loss < loss_tol, e_loss
    bad = {k: v for k, v in errs.items() if v > (stem_tol if k.startswith("patch_embed.") else grad_tol)}
    assert not bad, sorted(bad.items(), key=lambda kv: -kv[1])[:10]


def test_volo_d5_shapes_448_vs_oracle():
    """BASELINE configs[4] shapes in bf16: VOLO-D5 widths (384 / 768 channels, 12 / 16 heads -> head_dim 32 outlook, head_dim 48
    attention, mlp ratio 4, stem width 128; models/volo.py:799-821) at 448 px (56x56 outlook grid, 784 tokens -> the key/query-
    blocked attention kernels), one block per kind, batch 1: forward, loss and every parameter gradient against the oracle."""
    _d5_shapes_vs_oracle(out_tol=3e-2, loss_tol=3e-3, grad_tol=6e-2, stem_tol=0.14, tag="bf16")


@pytest.mark.parametrize("fp8_dgrad", [False, True])
def test_volo_d5_shapes_448_fp8_vs_oracle(monkeypatch, fp8_dgrad):
    """The same D5-shape network under functional.FP8_LINEAR (configs[4]: "mixed MFMA fp8 GEMM / bf16 accum"): the
    forward Linear products of its transformer blocks run on e4m3 operands at the D5 shapes (K = 768 / 3072, head_dim 48 blocked
    attention emitting its e4m3 output) -- compared with the fp64 ORACLE, not with the bf16 HIP path.  e4m3 keeps 3 mantissa bits
    (2^-4 relative per element, ~1/sqrt(K) of it after a K-long dot product), so the bounds are wider than the bf16 test's: outputs
    <= 8e-2, loss <= 1e-2 relative, parameter gradients <= 0.12 per tensor (0.2 in the conv stem, whose BatchNorm backward amplifies
    the perturbation of everything above it).  Measured (round 4): outputs 4.8e-2 / 6.0e-2, loss 4.0e-3, gradients median 3.6e-2, max
    0.103 (patch_embed.conv.1.weight); the bf16 run of the same network: 7.0e-3 / 8.4e-3, 7e-5, 7.4e-3, 9.8e-2.
    fp8_dgrad (round 5, functional.FP8_DGRAD): the INPUT-GRADIENT product of fc1 (K = 3072) on e4m3 operands as well -- dL/dh quantised with a
    per-tensor scale, the transposed weight with the forward copy's -- under the same bounds (the gradients are then fp8 GRADIENTS against the
    fp64 oracle)."""
    from autoprog_amd import functional as AF, ops
    AF.reset_fp8_state()
    monkeypatch.setattr(AF, "FP8_LINEAR", True)
    calls = []
    real = ops.gemm_nt_fp8
    monkeypatch.setattr(ops, "gemm_nt_fp8", lambda *a, **k: (calls.append(tuple(a[0].shape)), real(*a, **k))[1])
    monkeypatch.setattr(AF, "FP8_DGRAD", fp8_dgrad)
    try:
        _d5_shapes_vs_oracle(out_tol=8e-2, loss_tol=1e-2, grad_tol=0.12, stem_tol=0.2, tag="fp8 forward GEMMs" + (" + fc1 input gradient" if fp8_dgrad else ""))
    finally:
        AF.reset_fp8_state()
    # two transformer blocks x (qkv, proj, fc1, fc2) at 784 tokens: K = 768 (x3) and 3072; + the input-gradient product of fc1 (K = 3072)
    assert len(calls) == (10 if fp8_dgrad else 8) and sorted({s[1] for s in calls}) == [768, 3072], calls
    assert sum(1 for s in calls if s[1] == 3072) == (4 if fp8_dgrad else 2)


def test_volo_d5_full_depth_448_smoke():
    """volo_d5(img_size=448) (models/volo.py:799-821; BASELINE configs[4] in bf16): one full forward / loss / backward at batch 2
    through the HIP kernels -- shapes, finiteness, every parameter receives a gradient, elastic depth mask works on it."""
    from autoprog_amd.models import create_model
    from autoprog_amd.loss import TokenLabelCrossEntropy
    torch.manual_seed(0)
    model = create_model("volo_d5", img_size=448, drop_path_rate=0.1).cuda().train()
    assert sum(p.numel() for p in model.parameters()) == 295907168          # SURVEY.md / BASELINE.md: exact D5 parameter count
    x = torch.randn(2, 3, 448, 448, device="cuda")
    target = torch.softmax(torch.randn(2, 1000, 2 + 784, device="cuda") * 3, dim=1)
    np.random.seed(0)
    out = model(x)
    assert out[0].shape == (2, 1000) and out[1].shape == (2, 784, 1000)
    loss = TokenLabelCrossEntropy(dense_weight=0.5, cls_weight=1.0)(out, target)
    loss.backward()
    assert torch.isfinite(loss) and 6.0 < float(loss) < 12.0, float(loss)
    for n, p in model.named_parameters():
        assert p.grad is not None and torch.isfinite(p.grad).all(), n
    assert float(model.network[3][5].attn.qkv.weight.grad.abs().sum()) > 0


def test_product_never_imports_oracle():
    import os
    import re
    import autoprog_amd
    for root, _, files in os.walk(autoprog_amd.__path__[0]):
        for f in files:
            if f.endswith(".py"):
                text = open(os.path.join(root, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", text, re.M), (root, f)


def test_gradient_sink_matches_autograd_accumulation():
    """fused backward accumulating straight into the reducer's slab == autograd-returned gradients"""
    from autoprog_amd.loss import TokenLabelCrossEntropy
    from autoprog_amd.dist import GradientBucketReducer
    d = load("volo_full")
    tag = "h2_l6"
    x = torch.from_numpy(d[tag + ".x"]).cuda()
    target = torch.from_numpy(d[tag + ".target"]).cuda()
    loss_fn = TokenLabelCrossEntropy(dense_weight=0.5, cls_weight=1.0, classes=12)
    grads = []
    for use_sink in (False, True):
        model = load_sd(build("volo_h2_l6", 12), d, tag).cuda().train()
        reducer = GradientBucketReducer(list(model.parameters()), world_size=1) if use_sink else None
        if reducer:
            reducer.install_sink(model)
            reducer.zero_grad()
        np.random.seed(11)
        loss_fn(model(x), target).backward()
        if reducer:
            reducer.finish()
            reducer.remove()
        grads.append({n: p.grad.detach().clone() for n, p in model.named_parameters()})
    for n in grads[0]:
        assert rel(grads[1][n], grads[0][n]) < 1e-3, n


def test_deit_tiny_depth4_vs_oracle():
    """BASELINE.json configs[0] shape (DeiT-Tiny 224, depth-4 sub-net, soft-target CE) at batch 4:
    HIP path vs the oracle's timm-VisionTransformer restatement (head_dim 64 attention kernels)."""
    from autoprog_amd.models import create_model
    from autoprog_amd.loss import SoftTargetCrossEntropy
    torch.manual_seed(0)
    model = create_model("model_variant", variant="deit_h3_l4").cuda().train()
    B = 4
    x = torch.randn(B, 3, 224, 224, device="cuda")
    target = torch.softmax(torch.randn(B, 1000, device="cuda") * 3, dim=-1)
    y = model(x)
    loss = SoftTargetCrossEntropy()(y, target)
    loss.backward()
    p = {k: v.detach().double().cpu().requires_grad_(True) for k, v in model.state_dict().items()}
    yr = R.vit_forward(p, x.double().cpu(), depth=4, heads=3)
    lr = R.soft_target_ce(yr, target.double().cpu())
    lr.backward()
    assert rel(y, yr) < 2e-2, rel(y, yr)
    assert abs(float(loss.detach()) - float(lr.detach())) < 2e-3 * float(lr.detach())
    errs = {n: rel(q.grad, p[n].grad) for n, q in model.named_parameters() if float(p[n].grad.norm()) > 1e-9}
    bad = {k: v for k, v in errs.items() if v > 6e-2}
    assert not bad, sorted(bad.items(), key=lambda kv: -kv[1])[:8]
    # elastic depth: skipping block 1 == the oracle with that block removed
    model.eval()
    skip = model.set_sample_config(dict(layer_num=3, min_layer_num=2, max_layer_num=4))
    with torch.no_grad():
        ys = model(x)
    yrs = R.vit_forward({k: v.detach() for k, v in p.items()}, x.double().cpu(), depth=4, heads=3, train=False, skip=skip)
    assert rel(ys, yrs) < 2e-2


def test_deit_kernels_vs_fp64_with_the_same_rounding_points():
    """DeiT (timm VisionTransformer as models/deit.py registers it) against the oracle's ViT with the pipeline's bf16 rounding points
    (vit_forward(bf16_points=True)): DeiT-Tiny width, depth 4, 224 px, batch 4, soft-target CE -- outputs, loss, every parameter
    gradient.  Stage by stage (each fed from the oracle's tensor) the forward agrees to 2e-5 (patch projection), 6e-5 - 2.4e-4 (blocks) and
    exactly (position embedding, final norm, head); end to end on this default-initialised network (logits near zero) the output is
    6.5e-3 away, the loss 1.7e-5 relative, the gradients 6.2e-3 in the median and 8.9e-3 at worst.  Bound DEIT_KERNEL_TOL = 2e-2
    (the plain-oracle tests above: 2e-2 / 6e-2)."""
    from autoprog_amd.models import create_model
    from autoprog_amd.loss import SoftTargetCrossEntropy
    torch.manual_seed(0)
    model = create_model("model_variant", variant="deit_h3_l4").cuda().train()
    B = 4
    x = torch.randn(B, 3, 224, 224, device="cuda")
    target = torch.softmax(torch.randn(B, 1000, device="cuda") * 3, dim=-1)
    y = model(x)
    loss = SoftTargetCrossEntropy()(y, target)
    loss.backward()
    p = {k: v.detach().double().cpu().requires_grad_(True) for k, v in model.state_dict().items()}
    yr = R.vit_forward(p, x.double().cpu(), depth=4, heads=3, bf16_points=True)
    lr = R.soft_target_ce(yr, target.double().cpu())
    lr.backward()
    e_y = rel(y, yr)
    e_l = abs(float(loss.detach()) - float(lr.detach())) / float(lr.detach())
    errs = {n: rel(q.grad, p[n].grad) for n, q in model.named_parameters() if float(p[n].grad.norm()) > 1e-12}
    print("DeiT vs rounding-matched fp64: output %.2e, loss rel %.2e, gradients: median %.2e max %.2e (%s)"
          % (e_y, e_l, float(np.median(list(errs.values()))), max(errs.values()), max(errs, key=errs.get)))
    assert e_y < DEIT_KERNEL_TOL and e_l < 1e-4, (e_y, e_l)
    bad = {k: v for k, v in errs.items() if v > DEIT_KERNEL_TOL}
    assert not bad, sorted(bad.items(), key=lambda kv: -kv[1])[:8]


DEIT_KERNEL_TOL = 2e-2


def test_deit_distilled_contract():
    """DistilledVisionTransformer (models/deit.py:20-59): 198 tokens, train returns (x, x_dist), eval their mean; gradients
    reach the distillation token, its position embedding and head_dist."""
    from autoprog_amd.models import create_model
    torch.manual_seed(2)
    model = create_model("deit_tiny_distilled_patch16_224", num_classes=16).cuda().train()
    x = torch.randn(2, 3, 224, 224, device="cuda")
    y, yd = model(x)
    assert y.shape == (2, 16) and yd.shape == (2, 16) and model.pos_embed.shape[1] == 198
    (y.float().square().mean() + yd.float().square().mean()).backward()
    for n in ("dist_token", "pos_embed", "head_dist.weight", "head.weight", "blocks.0.attn.qkv.weight"):
        g = dict(model.named_parameters())[n].grad
        assert g is not None and torch.isfinite(g).all() and float(g.abs().sum()) > 0, n
    model.eval()
    with torch.no_grad():
        ye = model(x)
        model.train()
        y2, yd2 = model(x)                    # drop_path_rate 0: train and eval forward are the same function
    assert rel(ye, (y2.float() + yd2.float()) / 2) < 1e-2


def test_deit_base_width_vs_oracle():
    """BASELINE.json configs[3] kernel shapes (DeiT-Base width: 768 channels, 12 heads of 64, 197 tokens) on a 2-block
    network at batch 2: forward, loss and every parameter gradient vs the oracle."""
    from autoprog_amd.models import create_model
    from autoprog_amd.loss import SoftTargetCrossEntropy
    torch.manual_seed(1)
    model = create_model("model_variant", variant="deit_h12_l2").cuda().train()
    B = 2
    x = torch.randn(B, 3, 224, 224, device="cuda")
    target = torch.softmax(torch.randn(B, 1000, device="cuda") * 3, dim=-1)
    y = model(x)
    loss = SoftTargetCrossEntropy()(y, target)
    loss.backward()
    p = {k: v.detach().double().cpu().requires_grad_(True) for k, v in model.state_dict().items()}
    yr = R.vit_forward(p, x.double().cpu(), depth=2, heads=12)
    lr = R.soft_target_ce(yr, target.double().cpu())
    lr.backward()
    assert rel(y, yr) < 2e-2, rel(y, yr)
    assert abs(float(loss.detach()) - float(lr.detach())) < 2e-3 * float(lr.detach())
    errs = {n: rel(q.grad, p[n].grad) for n, q in model.named_parameters() if float(p[n].grad.norm()) > 1e-9}
    bad = {k: v for k, v in errs.items() if v > 6e-2}
    assert not bad, sorted(bad.items(), key=lambda kv: -kv[1])[:8]


def test_deit_distilled_vs_oracle():
    """DistilledVisionTransformer (models/deit.py:20-59) against the oracle's restatement: both heads, loss on their
    sum-of-CEs, every parameter gradient (dist_token, the 198-row pos_embed and head_dist included); eval = mean of heads."""
    from autoprog_amd.models import create_model
    from autoprog_amd.loss import SoftTargetCrossEntropy
    torch.manual_seed(4)
    model = create_model("deit_tiny_distilled_patch16_224", num_classes=40).cuda().train()
    model.blocks = model.blocks[:2]                                   # two blocks keep the oracle fast; same code path
    B = 3
    x = torch.randn(B, 3, 224, 224, device="cuda")
    target = torch.softmax(torch.randn(B, 40, device="cuda") * 3, dim=-1)
    y, yd = model(x)
    ce = SoftTargetCrossEntropy()
    loss = ce(y, target) + 0.5 * ce(yd, target)
    loss.backward()
    p = {k: v.detach().double().cpu().requires_grad_(True) for k, v in model.state_dict().items()}
    yr, ydr = R.vit_forward(p, x.double().cpu(), depth=2, heads=3, distilled=True)
    lr = R.soft_target_ce(yr, target.double().cpu()) + 0.5 * R.soft_target_ce(ydr, target.double().cpu())
    lr.backward()
    assert rel(y, yr) < 2e-2 and rel(yd, ydr) < 2e-2, (rel(y, yr), rel(yd, ydr))
    assert abs(float(loss.detach()) - float(lr.detach())) < 2e-3 * float(lr.detach())
    errs = {n: rel(q.grad, p[n].grad) for n, q in model.named_parameters() if float(p[n].grad.norm()) > 1e-9}
    assert {"dist_token", "pos_embed", "head_dist.weight"} <= set(errs)
    bad = {k: v for k, v in errs.items() if v > 6e-2}
    assert not bad, sorted(bad.items(), key=lambda kv: -kv[1])[:8]
    model.eval()
    with torch.no_grad():
        ye = model(x)
    yre = R.vit_forward({k: v.detach() for k, v in p.items()}, x.double().cpu(), depth=2, heads=3, distilled=True, train=False)
    assert rel(ye, yre) < 2e-2


@pytest.mark.parametrize("r", [128, 160, 192])
def test_deit_elastic_resolution_vs_oracle(r):
    """BASELINE configs[3] elastic r in {128..224} on DeiT: the patch-grid part of pos_embed is resized per step (bicubic, VOLO's
    rule); forward + pos_embed / cls_token gradients vs the oracle at 64-wide heads, N = (r/16)^2 + 1 tokens."""
    from autoprog_amd.models import create_model
    torch.manual_seed(5)
    model = create_model("model_variant", variant="deit_h3_l2", num_classes=24).cuda().train()
    B = 2
    x = torch.randn(B, 3, r, r, device="cuda")
    y = model(x)
    y.float().square().mean().backward()
    p = {k: v.detach().double().cpu().requires_grad_(True) for k, v in model.state_dict().items()}
    yr = R.vit_forward(p, x.double().cpu(), depth=2, heads=3)
    yr.square().mean().backward()
    assert rel(y, yr) < 2e-2, rel(y, yr)
    for n in ("pos_embed", "cls_token", "blocks.0.attn.qkv.weight", "patch_embed.proj.weight"):
        e = rel(dict(model.named_parameters())[n].grad, p[n].grad)
        assert e < 6e-2, (n, e)


@pytest.mark.parametrize("l,r", [(9, 128), (12, 160), (15, 192)])
def test_autoprog_stage_shapes_elastic_supernet(l, r):
    """BASELINE.json configs[2]: the reference schedule's stage shapes (l, r) run on ONE volo_h12_l18
    supernet through set_sample_config (elastic depth as a launch-time mask) and a resized input
    (elastic token count + bicubic pos-embed interpolation), against the oracle with the same skip table."""
    from autoprog_amd.models import create_model
    torch.manual_seed(1)
    model = create_model("model_variant", variant="volo_h12_l18", num_classes=1000, img_size=224).cuda().train()
    mask = model.set_sample_config(dict(layer_num=l, min_layer_num=9, max_layer_num=18))
    B = 2
    x = torch.randn(B, 3, r, r, device="cuda")
    np.random.seed(l)
    x_cls, x_aux, bb = model(x)
    assert x_aux.shape == (B, (r // 16) ** 2, 1000)
    p = {k: v.detach().double().cpu() for k, v in model.state_dict().items()}
    rng = np.random.RandomState(l)
    lam, box = R.draw_mix_box((B, r // 8, r // 8, 192), 2, 1.0, rng)
    assert tuple(bb) == tuple(box)
    skip = R.skip_layer_table(l, 9, 18)
    assert [sorted(s) for s in mask.skip] == [sorted(s) for s in skip]
    arch = R.variant_arch("volo_h12_l18")
    ref_cls, ref_aux, _ = R.volo_forward(p, x.double().cpu(), train=True, mix=(lam, box), skip=skip, **arch)
    assert rel(x_cls, ref_cls) < 3e-2 and rel(x_aux, ref_aux) < 3e-2
    (x_cls.float().sum() + x_aux.float().mean()).backward()
    active = model.network[2][0].attn.qkv.weight.grad
    skipped = model.network[2][sorted(skip[1])[0]].attn.qkv.weight.grad if skip[1] else None
    assert active is not None and float(active.abs().sum()) > 0
    assert skipped is None or float(skipped.abs().sum()) == 0.0          # identity layers receive no gradient


def test_loss_curve_with_fused_optimizer_and_sink():
    """same 5-step reference curve, but through the production step: gradient sink + fused AdamW/EMA kernel that
    also refreshes the bf16 weight copies (a stale copy would freeze the loss)"""
    from autoprog_amd.loss import TokenLabelCrossEntropy
    from autoprog_amd.dist import GradientBucketReducer
    from autoprog_amd.optim import FlatAdamWEma
    d = load("step_curve")
    model = build("volo_h2_l3", 16)
    model.load_state_dict({k[2:]: torch.from_numpy(np.asarray(v)) for k, v in d.items() if k.startswith("w.")})
    model = model.cuda().train()
    x = torch.from_numpy(d["x"]).cuda()
    target = torch.from_numpy(d["target"]).cuda()
    red = GradientBucketReducer(list(model.parameters()), world_size=1)
    red.install_sink(model)
    opt = FlatAdamWEma(model, red, lr=float(d["lr"]), weight_decay=float(d["wd"]), ema_decays=[0.9, 0.99])
    loss_fn = TokenLabelCrossEntropy(dense_weight=0.5, cls_weight=1.0, classes=16)
    np.random.seed(int(d["np_seed"]))
    losses = []
    try:
        for step in range(5):
            red.zero_grad()
            loss = loss_fn(model(x), target)
            loss.backward()
            red.finish()
            opt.step()
            losses.append(float(loss.detach()))
    finally:
        red.remove()
    diff = np.abs(np.array(losses) - d["losses"]) / d["losses"]
    assert diff.max() < 4e-3, (losses, d["losses"].tolist())          # measured 1.0e-3 - 2.2e-3 relative (round 2 and 3 runs)
    ema = opt.ema_state_dict(0)
    w = dict(model.named_parameters())["head.weight"]
    assert not torch.equal(ema["head.weight"], w.detach()) and torch.isfinite(ema["head.weight"]).all()


def _realistic_init_setup():
    from tests._initweights import init_state_dict
    d = load("step_curve_init")
    classes = int(d["classes"])
    model = build("volo_h4_l6", classes)
    model.load_state_dict(init_state_dict(model.state_dict(), int(d["init_seed"])), strict=True)
    return d, classes, model.cuda().train()


def _param_groups(model, wd):
    decay, no_decay = [], []
    for n, p in model.named_parameters():
        (no_decay if (p.dim() == 1 or n.endswith(".bias") or n in ("pos_embed", "cls_token")) else decay).append(p)
    return [{"params": decay, "weight_decay": wd}, {"params": no_decay, "weight_decay": 0.0}]


def test_loss_curve_realistic_init_vs_reference():
    """north_star / SURVEY O5 "loss-curve parity 1e-3" on the realistic-init fixture (reference volo_h4_l6 run in fp64,
    trunc-normal .02 weights, 10 AdamW steps, tests/golden/step_curve_init.npz).

    What is held, and why not 1e-3 on all ten steps: AdamW's first updates are ~lr*sign(g), so ANY perturbation of a
    small gradient element becomes an O(lr) weight difference and this curve amplifies rounding chaotically from step 4 on.
    The fixture records the reference's OWN deviations from its exact (fp64) curve as yardsticks: plain fp32 7.7e-4; under
    torch.autocast fp16 (the stand-in for its apex-O1 training path) 2.3e-3; under torch.autocast bf16 -- the arithmetic class
    of this implementation -- 4.7e-2 already at step 0 and 0.44 at step 7.  So 1e-3 is not met by the reference's own 16-bit
    paths either.  The HIP path was measured at 2.5e-4 (step 0), 1.6e-3 (steps 0-3) and 1.9e-2 (worst, step 4) with the split
    class blocks, and at 3e-5 / 4e-4 / 4.4e-3 with the concatenating ones: two arrangements of the SAME arithmetic whose
    per-block errors against fp64 are equal (tools/exp_clsblock.py), i.e. the spread between them is the chaos of the curve,
    not an error of either.  Asserted: step 0 <= 1e-3, steps 0-3 <= 2.5e-3, every step <= 2.5e-2 AND <= a tenth of the
    reference's own bf16-autocast deviation; first-step gradients <= 2.5e-2 rel-L2 per tensor outside the conv stem and <= 0.1
    inside it (BatchNorm backward + bf16 activations: heavy cancellation), gradient norms <= 6e-2."""
    from autoprog_amd.loss import TokenLabelCrossEntropy
    d, classes, model = _realistic_init_setup()
    x = torch.from_numpy(d["x"]).cuda()
    target = torch.from_numpy(d["target"]).cuda()
    opt = torch.optim.AdamW(_param_groups(model, float(d["wd"])), lr=float(d["lr"]))
    loss_fn = TokenLabelCrossEntropy(dense_weight=0.5, cls_weight=1.0, classes=classes)
    np.random.seed(int(d["np_seed"]))
    losses = []
    for step in range(10):
        out = model(x)
        assert [int(v) for v in out[2]] == [int(v) for v in d["boxes"][step]]
        loss = loss_fn(out, target)
        opt.zero_grad()
        loss.backward()
        if step == 0:
            named = dict(model.named_parameters())
            errs = {k[3:]: rel(named[k[3:]].grad, v) for k, v in d.items() if k.startswith("g0.")}
            norms = {str(n): abs(float(named[str(n)].grad.double().norm()) - float(g)) / float(g)
                     for n, g in zip(d["g0_norms_names"], d["g0_norms"]) if float(g) > 1e-12}
            print("realistic-init first-step gradient rel-L2 errors:", {k: round(v, 4) for k, v in sorted(errs.items(), key=lambda kv: -kv[1])[:6]})
            print("worst gradient-norm deviations:", {k: round(v, 4) for k, v in sorted(norms.items(), key=lambda kv: -kv[1])[:6]})
        opt.step()
        losses.append(float(loss.detach()))
    dev = np.abs(np.array(losses) - d["losses"])
    print("realistic-init loss curve |hip - ref(fp64)|:", [round(float(v), 5) for v in dev],
          "\n   reference fp32 vs fp64:", [round(float(v), 5) for v in np.abs(d["losses_fp32"] - d["losses"])],
          "\n   reference fp16-autocast vs fp64:", [round(float(v), 5) for v in np.abs(d["losses_fp16_autocast"] - d["losses"])],
          "\n   reference bf16-autocast vs fp64:", [round(float(v), 5) for v in np.abs(d["losses_bf16_autocast"] - d["losses"])])
    bad = {k: v for k, v in errs.items() if v > (0.1 if k.startswith("patch_embed.conv") else 2.5e-2)}
    assert not bad, bad
    assert max(norms.values()) < 6e-2, max(norms.values())
    # measured (round 3): 2.3e-4 at step 0, <= 2.6e-4 over steps 0-3, 4.5e-3 worst (step 8); round 2 on another kernel arrangement
    # 2.5e-4 / 1.6e-3 / 1.9e-2: the tail is the chaos described above, the head is held at north_star's 1e-3
    assert dev[0] < 1e-3 and dev[:4].max() < 2e-3, dev.tolist()
    assert dev.max() < 2.5e-2, (losses, d["losses"].tolist())
    assert dev.max() < 0.1 * np.abs(d["losses_bf16_autocast"] - d["losses"]).max()


def test_late_state_vs_reference_golden(monkeypatch):
    """Parity on a TRAINED network (Weak 2; tests/golden/late_state.npz, tools/gen_golden.py::gen_late_state): the
    reference's volo_h4_l6 after 300 fp64 AdamW steps on its batch (loss 5.4 -> 3.0), its weights as fp32, evaluated by the reference in
    fp64 on a fresh mix-token draw.  The HIP model loads those weights and runs the same step.  Round 5: the fixture's network has the
    SHIPPED 64-wide stem (tools/gen_golden.py LATE_STEM; no torch convolution may run).  Asserted (bounds next to the measured values):
    logits <= 2e-2 rel-L2, loss <= 1e-2 relative, all parameter gradients as ONE vector <= 0.1 rel-L2, no tensor above 0.2 -- the stem
    included (rounds 3 - 4, 16-wide MIOpen stem: 0.25 / 0.5 inside the stem).  The fixture also records the reference's own network
    under torch.autocast("cpu", bfloat16) on these weights: logits 2.1e-3 / 4.0e-3, gradients 7.4e-2 as one vector, median 5.9e-2, worst
    0.119 -- with the 64-wide stem that recipe survives this state and IS a yardstick: the HIP kernels sit at 7.6e-2 / 6.0e-2 / 0.137."""
    from autoprog_amd.loss import TokenLabelCrossEntropy
    forbid_torch_convolutions(monkeypatch)            # round 5: the fixture runs on the shipped 64-wide stem (tools/gen_golden.py LATE_STEM)
    d = load("late_state")
    classes = int(d["classes"])
    model = build("volo_h4_l6", classes, stem=64)
    sd = {k[2:]: torch.from_numpy(np.asarray(v)) for k, v in d.items() if k.startswith("w.")}
    model.load_state_dict(sd, strict=True)
    model = model.cuda().train()
    x = torch.from_numpy(d["x"]).cuda()
    target = torch.from_numpy(d["target"]).cuda()
    np.random.seed(int(d["np_seed"]))
    out = model(x)
    assert [int(v) for v in out[2]] == [int(v) for v in d["box"]]
    e_cls, e_aux = rel(out[0], d["y_cls"]), rel(out[1], d["y_aux"])
    loss = TokenLabelCrossEntropy(dense_weight=0.5, cls_weight=1.0, classes=classes)(out, target)
    loss.backward()
    named = dict(model.named_parameters())
    names = [str(n) for n in d["grad_names"]]
    ref = {k: torch.from_numpy(d["g16." + k].astype(np.float64)) * float(d["gs." + k]) for k in names}
    errs = {k: rel(named[k].grad, ref[k]) for k in names}
    va = torch.cat([named[k].grad.detach().double().cpu().flatten() for k in names])
    vb = torch.cat([ref[k].flatten() for k in names])
    one = float((va - vb).norm() / vb.norm())
    worst = sorted(errs.items(), key=lambda kv: -kv[1])[:8]
    print("late state: logits %.4f / %.4f, loss %.5f (reference %.5f), gradients: one vector %.4f, median %.4f, worst %s"
          % (e_cls, e_aux, float(loss), float(d["loss"]), one, float(np.median(list(errs.values()))), [(k, round(v, 4)) for k, v in worst]))
    assert e_cls < 2e-2 and e_aux < 2e-2, (e_cls, e_aux)
    assert abs(float(loss) - float(d["loss"])) < 1e-2 * float(d["loss"])
    # measured (round 5, 64-wide HIP stem): logits 2.1e-3 / 4.9e-3, loss 3.29509 against 3.29513, gradients: one vector 7.6e-2, median
    # 6.0e-2, worst 0.137 / 0.121 / 0.121 (stem convolution weights) -- in this memorising state the gradients themselves are small
    # differences of large terms, which is why they sit an order of magnitude above the first-step gradients of the same network
    # (round 4, 16-wide MIOpen stem: 7.8e-2 / 5.7e-2 / 0.20)
    assert one < 0.1, one
    yard = float(d["yard_bf16_autocast_one_vector"])
    assert one < 1.25 * yard, (one, yard)        # within a quarter of the reference's own bf16-autocast recipe on the same state
    bad = {k: v for k, v in errs.items() if v > 0.2}
    assert not bad, bad
    # The same step against the oracle WITH the pipeline's bf16 rounding points (volo_forward(bf16_points=True)): the yardstick the
    # second yardstick.  Measured (round 5): that oracle -- exact arithmetic between the same roundings -- is itself 7.7e-2 away from
    # the fp64 reference as one gradient vector; the HIP kernels 7.6e-2 from the reference and 6.5e-2 from that oracle (logits 2.0e-3 /
    # 2.8e-3, loss 3.29509 / 3.29514): in this memorising state ANY two evaluations that round to bf16 part by ~7 %, the gradients
    # being small differences of large terms -- the kernels sit where an exact implementation of the 16-bit recipe sits.
    p64 = {k: v.double().clone() for k, v in sd.items()}
    for v in p64.values():
        if v.dtype.is_floating_point:
            v.requires_grad_(True)
    box = tuple(int(v) for v in d["box"])
    ref2 = R.volo_forward(p64, x.double().cpu(), train=True, mix=(1.0, box), bf16_points=True, **R.variant_arch("volo_h4_l6"))
    loss2 = R.token_label_ce(ref2, target.double().cpu(), 0.5, 1.0)
    loss2.backward()
    e2 = (rel(out[0], ref2[0]), rel(out[1], ref2[1]))
    errs2 = {k: rel(named[k].grad, p64[k].grad) for k in names}
    vc = torch.cat([p64[k].grad.flatten() for k in names])
    one2 = float((va - vc).norm() / vc.norm())
    recipe = float((vc - vb).norm() / vb.norm())
    print("late state vs the rounding-matched oracle: logits %.4f / %.4f, loss %.5f (oracle %.5f), gradients: one vector %.4f, median %.4f, worst %s; "
          "the rounding-matched oracle against the fp64 reference, one vector: %.4f"
          % (e2[0], e2[1], float(loss), float(loss2), one2, float(np.median(list(errs2.values()))),
             [(k, round(v, 4)) for k, v in sorted(errs2.items(), key=lambda kv: -kv[1])[:4]], recipe))
    assert max(e2) < LATE_POINTS_OUT_TOL and one2 < LATE_POINTS_GRAD_TOL, (e2, one2)


LATE_POINTS_OUT_TOL, LATE_POINTS_GRAD_TOL = 2e-2, 0.1


def test_hip_stem_eval_and_elastic_resolution_vs_oracle():
    """The 64-wide HIP stem (conv7 on the space-to-depth image, the two 3x3 convolutions, patch-addressed proj; models/volo.py:342-380)
    in EVAL mode (running statistics) and with the per-step input resize (main_prog.py:973-974: a 96 px batch fed to a model set to
    64 px) against the oracle on the resized image.  Outputs <= 3e-2 rel-L2 (bf16 network, as the other full-network checks)."""
    import torch.nn.functional as F
    from autoprog_amd.models import create_model
    torch.manual_seed(3)
    model = create_model("model_variant", variant="volo_h12_l9", num_classes=40, img_size=224).cuda()
    with torch.no_grad():                                  # non-trivial running statistics
        for m in model.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.normal_(0, 0.2); m.running_var.uniform_(0.5, 1.5)
    model.eval()
    arch = R.variant_arch("volo_h12_l9")
    p = {k: v.detach().double().cpu() for k, v in model.state_dict().items()}
    x = torch.randn(2, 3, 96, 96, device="cuda")
    model.set_sample_config(dict(layer_num=9, min_layer_num=9, max_layer_num=9, input_size=64, token_label_size=4))
    model.patch_embed.hip_conv = True          # (AP_STEM_HIP_CONV=0 runs of the suite flip the default)
    # the stage resolution belongs to training: an eval() forward keeps its own resolution unless asked (reference: main_prog.py:973
    # resizes in the training loop only)
    with torch.no_grad():
        y96 = model(x)
    ref96 = R.volo_forward(p, x.double().cpu(), train=False, **arch)
    assert rel(y96, ref96) < 3e-2, rel(y96, ref96)
    model.patch_embed.resize_in_eval = True
    with torch.no_grad():
        y = model(x)
    xr = F.interpolate(x.double().cpu(), size=(64, 64), mode="bilinear", align_corners=False)
    ref = R.volo_forward(p, xr, train=False, **arch)
    assert rel(y, ref) < 3e-2, rel(y, ref)
    # the same input through the MIOpen stem: the two stems agree to bf16 accuracy
    model.patch_embed.hip_conv = False
    with torch.no_grad():
        y2 = model(x)
    assert rel(y, y2) < 2e-2, rel(y, y2)


@pytest.mark.parametrize("fp8_dgrad", [False, True])
def test_fp8_forward_gemms_inside_the_transformer_block(monkeypatch, fp8_dgrad):
    """BASELINE configs[4] ('mixed MFMA fp8 GEMM'): with functional.FP8_LINEAR the four Linear layers of a transformer block run their
    forward product on e4m3 operands -- the fp8 instantiation of the 8-phase kernel at this size (M = 4096, K % 128 == 0) -- with
    per-tensor delayed scaling; the backward stays bf16 (fp8_dgrad = False) or runs fc1's input-gradient product on e4m3 operands too
    (functional.FP8_DGRAD, round 5: in the first step dL/dh is quantised by a pass of its own, in the second by the epilogue of the launch
    that produces it -- the mul_by8 + q8 flavour of the bf16 8-phase kernel).  Against the bf16 block on the same weights and input: output <= 3e-2,
    weight gradients <= 8e-2 (e4m3 keeps 3 mantissa bits; the residual stream is not quantised).  The second step quantises with the
    scales rolled from the first step's amax (no reduction pass in front of the quantiser) and stays as close."""
    from autoprog_amd import functional as AF, ops
    from autoprog_amd.models.volo import Transformer
    AF.reset_fp8_state()
    monkeypatch.setattr(AF, "FP8_DGRAD", fp8_dgrad)
    torch.manual_seed(0)
    B, N, C, heads = 16, 256, 256, 8
    blk = Transformer(C, heads, mlp_ratio=3.0).cuda().train()
    x = torch.randn(B, 16, 16, C, device="cuda").to(torch.bfloat16)
    dy = torch.randn_like(x)

    def run(fp8):
        monkeypatch.setattr(AF, "FP8_LINEAR", fp8)
        blk.zero_grad()
        xi = x.clone().requires_grad_(True)
        y = blk(xi)
        y.backward(dy)
        return y.detach().float(), {n: p.grad.detach().clone() for n, p in blk.named_parameters()}, xi.grad.detach().float()

    calls = []
    real = ops.gemm_nt_fp8
    monkeypatch.setattr(ops, "gemm_nt_fp8", lambda *a, **k: (calls.append(a[0].shape), real(*a, **k))[1])
    y16, g16, dx16 = run(False)
    assert not calls
    y8, g8, dx8 = run(True)
    assert len(calls) == (5 if fp8_dgrad else 4) and all(s[0] == B * N for s in calls)
    rel = lambda a, b: float((a.float() - b.float()).norm() / b.float().norm())
    e_y = rel(y8, y16)
    e_g = {n: rel(g8[n], g16[n]) for n in g16}
    print("fp8 forward vs bf16 block: y %.4f, dx %.4f, grads %s" % (e_y, rel(dx8, dx16), {k: round(v, 4) for k, v in e_g.items()}))
    assert 0 < e_y < 3e-2 and rel(dx8, dx16) < 8e-2 and max(e_g.values()) < 8e-2
    # delayed scaling: an optimizer step later the scales come from the recorded amax, the LayerNorms emit the e4m3 operand of qkv /
    # fc1 themselves and the four weights are re-quantised in one launch
    sc = AF.fp8_scales
    i = sc.slots[("x", id(blk.mlp.fc2.weight))]
    amax_seen = float(sc.amax[i])
    assert amax_seen > 0
    AF._WeightBank.generation += 1
    emitted = []
    real_nt = ops.gemm_nt
    monkeypatch.setattr(ops, "gemm_nt", lambda *a, **k: (emitted.append(k.get("q8") is not None), real_nt(*a, **k))[1])
    y8b, g8b, _ = run(True)
    assert any(emitted) == fp8_dgrad                     # dL/dh left its launch as e4m3 (no quantisation pass) iff the fp8 input gradient is on
    assert abs(float(sc.scale[i]) - ops.FP8_MAX / amax_seen) < 1e-3 * float(sc.scale[i]) and float(sc.amax[i]) > 0
    assert rel(y8b, y16) < 3e-2 and max(rel(g8b[n], g16[n]) for n in g16) < 8e-2


def test_fp8_training_steps_on_a_small_volo(monkeypatch):
    """the fp8 forward path through a whole model and optimizer steps (weights re-quantised after every step from the bf16 copies the
    fused optimizer maintains): the loss of the first step is within 2 % of the bf16 model's and training goes on"""
    from autoprog_amd import functional as AF
    from autoprog_amd.dist import GradientBucketReducer
    from autoprog_amd.loss import TokenLabelCrossEntropy
    from autoprog_amd.models import create_model
    from autoprog_amd.optim import FlatAdamWEma
    losses = {}
    for fp8 in (False, True):
        AF.reset_fp8_state()
        monkeypatch.setattr(AF, "FP8_LINEAR", fp8)
        torch.manual_seed(0); np.random.seed(0)
        model = create_model("model_variant", variant="volo_h2_l3", num_classes=16, img_size=64, stem_hidden_dim=16).cuda().train()
        red = GradientBucketReducer(list(model.parameters()), world_size=1)
        red.install_sink(model)
        opt = FlatAdamWEma(model, red, lr=1e-3, weight_decay=0.05, ema_decays=[0.9])
        loss_fn = TokenLabelCrossEntropy(dense_weight=0.5, cls_weight=1.0, classes=16)
        g = torch.Generator().manual_seed(1)
        x = torch.randn(4, 3, 64, 64, generator=g).cuda()
        target = torch.softmax(torch.randn(4, 16, 18, generator=g) * 2, dim=1).cuda()
        ls = []
        try:
            for _ in range(6):
                red.zero_grad()
                loss = loss_fn(model(x), target)
                loss.backward()
                red.finish()
                opt.step()
                ls.append(float(loss.detach()))
        finally:
            red.remove()
        losses[fp8] = ls
    print("bf16 losses", [round(v, 4) for v in losses[False]], "fp8 losses", [round(v, 4) for v in losses[True]])
    assert abs(losses[True][0] - losses[False][0]) < 2e-2 * losses[False][0]
    assert all(np.isfinite(losses[True])) and losses[True][-1] < losses[True][0]
