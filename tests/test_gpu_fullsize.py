"""Parity at BASELINE.json's FULL sizes (VOLO-D1, 224 px, per-GPU batch 128), where the CPU oracle would take minutes:
size-independent properties of each hot-path kernel, all through the C ABI -- linearity of the GEMMs, grouped == one-by-one
weight gradients, zero row sums of the cross-entropy gradient, adjointness of the outlook gather pair, batch-permutation
equivariance of attention, shift invariance of LayerNorm -- plus spot checks of random rows against fp64 torch."""
import pytest
import torch

pytestmark = pytest.mark.gpu

B, T1, T2 = 128, 128 * 28 * 28, 128 * 14 * 14          # images, outlooker-stage tokens, transformer-stage tokens


@pytest.fixture(scope="module")
def ops():
    from autoprog_amd import ops as o
    return o


def bf(*shape, seed=0, scale=1.0):
    g = torch.Generator(device="cuda").manual_seed(seed)
    return (torch.randn(*shape, device="cuda", generator=g) * scale).bfloat16()


def rel(a, b):
    a, b = a.detach().double(), b.detach().double().to(a.device)
    return float((a - b).norm() / (b.norm() + 1e-30))


def test_gemm_nt_linearity_and_row_samples(ops):
    for M, N, K in [(T2, 1152, 384), (T2, 384, 1152), (T1, 576, 192), (T1, 192, 576)]:
        a1, a2 = bf(M, K, seed=1), bf(M, K, seed=2)
        w = bf(N, K, seed=3, scale=K ** -0.5)
        y1, y2 = ops.gemm_nt(a1, w).float(), ops.gemm_nt(a2, w).float()
        y12 = ops.gemm_nt((a1.float() + a2.float()).bfloat16(), w).float()
        assert rel(y12, y1 + y2) < 1.5e-2, (M, N, K)                      # bf16 rounding of the summed operand and of three outputs
        rows = torch.randint(0, M, (64,), device="cuda")
        ref = a1[rows].double() @ w.double().t()
        assert rel(y1[rows], ref) < 1e-2, (M, N, K)


def test_gemm_tn_grouped_equals_single_at_block_size(ops):
    shapes = [(T2, 1152, 384), (T2, 384, 384), (T2, 1152, 384), (T2, 384, 1152)]
    probs, singles = [], []
    for i, (M, N1, N2) in enumerate(shapes):
        a, b = bf(M, N1, seed=10 + i), bf(M, N2, seed=20 + i)
        c, cs = torch.zeros(N1, N2, device="cuda"), torch.zeros(N1, device="cuda")
        probs.append((a, b, c, N1, N2, cs))
        c1, cs1 = torch.zeros(N1, N2, device="cuda"), torch.zeros(N1, device="cuda")
        ops.gemm_tn_acc(a, b, c1, colsum=cs1)
        singles.append((c1, cs1))
    ops.gemm_tn_acc_grouped(probs)
    for (a, b, c, n1, n2, cs), (c1, cs1) in zip(probs, singles):
        assert rel(c, c1) < 1e-4 and rel(cs, cs1) < 1e-4                  # same products, different fp32 atomic order
        cols = torch.randint(0, n2, (8,), device="cuda")
        assert rel(c[:, cols], a.double().t() @ b[:, cols].double()) < 3e-3


def test_soft_ce_gradient_rows_sum_to_zero(ops):
    M, C, N = T2, 1000, 196
    logits = bf(M, C, seed=5, scale=2.0)
    g = torch.Generator(device="cuda").manual_seed(6)
    target = torch.rand(B, C, 2 + N, device="cuda", generator=g)
    target[:, :, 2:] /= target[:, :, 2:].sum(1, keepdim=True)              # every token's soft label sums to one
    from autoprog_amd.functional import SoftTargetCEFn
    x = logits.clone().requires_grad_(True)
    view = target[:, :, 2:]
    loss = SoftTargetCEFn.apply(x, view, view.stride(0), view.stride(1), view.stride(2), N)
    loss.backward()
    d = x.grad.float()
    assert float(loss.detach()) > 0 and torch.isfinite(d).all()
    assert float(d.sum(1).abs().max()) < 2e-2 * float(d.abs().sum(1).max())   # softmax*1 - t sums to zero per row (bf16 rounding)
    rows = torch.randint(0, M, (32,), device="cuda")
    t_rows = torch.stack([target[r // N, :, 2 + r % N] for r in rows.tolist()]).double()
    ref_rows = -(t_rows * torch.log_softmax(logits[rows].double(), dim=-1)).sum(-1)
    xr = logits[rows].double().requires_grad_(True)
    (-(t_rows * torch.log_softmax(xr, dim=-1)).sum(-1)).sum().backward()
    assert rel(d[rows] * M, xr.grad) < 1e-2                                # the kernel's gradient is of the MEAN over rows
    assert float(ref_rows.mean()) > 0


def test_outlook_gather_pair_is_adjoint(ops):
    heads, H = 6, 28
    v, dy = bf(B, H, H, heads * 32, seed=7), bf(B, H, H, heads * 32, seed=8)
    logits = bf(B * 14 * 14, ops.round_up(heads * 81, 8), seed=9)
    y = ops.outlook_fwd(v, logits, heads, 32 ** -0.5)
    dv, dlogits = ops.outlook_bwd(v, logits, dy, heads, 32 ** -0.5)
    lhs = float((y.double() * dy.double()).sum())
    rhs = float((v.double() * dv.double()).sum())
    assert abs(lhs - rhs) < 2e-2 * max(abs(lhs), abs(rhs), float(y.double().norm() * dy.double().norm()) * 1e-2)
    # softmax backward: every 9-slot row of the logit gradient sums to zero
    dl = dlogits[:, :heads * 81].float().reshape(-1, heads, 9, 9)
    assert float(dl.sum(-1).abs().max()) < 2e-2 * float(dl.abs().sum(-1).max())


def test_mhsa_is_batch_permutation_equivariant(ops):
    heads, N, C = 12, 196, 384
    qkv = bf(B * N, 3 * C, seed=11)
    do = bf(B * N, C, seed=12)
    perm = torch.randperm(B, device="cuda")
    idx = (perm[:, None] * N + torch.arange(N, device="cuda")[None]).reshape(-1)
    o, lse = ops.mhsa_fwd(qkv, B, N, heads, 32 ** -0.5)
    o2, lse2 = ops.mhsa_fwd(qkv[idx].contiguous(), B, N, heads, 32 ** -0.5)
    assert torch.equal(o[idx], o2)                                       # images are independent workgroups: bit exact
    dq = ops.mhsa_bwd(qkv, o, do, lse, B, N, heads, 32 ** -0.5)
    dq2 = ops.mhsa_bwd(qkv[idx].contiguous(), o2, do[idx].contiguous(), lse2, B, N, heads, 32 ** -0.5)
    assert torch.equal(dq[idx], dq2)
    # attention rows are convex combinations of V: outputs stay inside the per-head min/max of v
    vv = qkv[:, 2 * C:].float().reshape(B, N, heads, 32)
    oo = o.float().reshape(B, N, heads, 32)
    assert bool((oo <= vv.amax(1, keepdim=True) + 2e-2).all()) and bool((oo >= vv.amin(1, keepdim=True) - 2e-2).all())


def test_layernorm_shift_invariance_and_samples(ops):
    for T, C in [(T2, 384), (T1, 192)]:
        x = bf(T, C, seed=13, scale=2.0)
        g = torch.rand(C, device="cuda") + 0.5
        b = torch.randn(C, device="cuda") * 0.1
        y, mean, rstd = ops.layernorm_fwd(x, g, b, 1e-5)
        y2, mean2, rstd2 = ops.layernorm_fwd((x.float() + 4.0).bfloat16(), g, b, 1e-5)
        assert rel(mean2 - 4.0, mean) < 3e-2 and rel(y2, y) < 3e-2          # the shifted input is re-rounded to bf16
        rows = torch.randint(0, T, (64,), device="cuda")
        ref = torch.nn.functional.layer_norm(x[rows].double(), (C,), g.double(), b.double(), 1e-5)
        assert rel(y[rows], ref) < 1e-2


def test_d1_b128_forward_vs_oracle_on_an_image_slice():
    """End to end at the BASELINE size: volo_h12_l18 (VOLO-D1), 224 px, batch 128, eval mode (running BatchNorm statistics, so the
    samples are independent) -- the fused logits of 8 of the 128 images against the oracle run on those 8 images alone.  Then one
    full-size training step on the same batch: finite loss near ln(1000) and finite, non-zero gradients in every parameter."""
    import numpy as np
    from oracle import ref_cpu as R
    from autoprog_amd.models import create_model
    from autoprog_amd.loss import TokenLabelCrossEntropy
    torch.manual_seed(11)
    model = create_model("model_variant", variant="volo_h12_l18", num_classes=1000, img_size=224, drop_path_rate=0.1).cuda()
    with torch.no_grad():
        for m in model.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.normal_(0, 0.1); m.running_var.uniform_(0.8, 1.2)
    x = torch.randn(128, 3, 224, 224, device="cuda")
    model.eval()
    with torch.no_grad():
        y = model(x)
    idx = [0, 17, 33, 64, 65, 99, 126, 127]
    p = {k: v.detach().double().cpu() for k, v in model.state_dict().items()}
    ref = R.volo_forward(p, x[idx].double().cpu(), train=False, **R.variant_arch("volo_h12_l18"))
    err = float((y[idx].double().cpu() - ref).norm() / ref.norm())
    assert err < 3e-2, err
    model.train()
    g = torch.Generator().manual_seed(1)
    target = torch.softmax(torch.randn(128, 1000, 198, generator=g) * 3, dim=1).cuda()
    np.random.seed(0)
    loss = TokenLabelCrossEntropy(dense_weight=0.5, cls_weight=1.0, classes=1000)(model(x), target)
    loss.backward()
    assert torch.isfinite(loss) and 6.0 < float(loss) < 14.0, float(loss)
    for n, q in model.named_parameters():
        assert q.grad is not None and torch.isfinite(q.grad).all() and float(q.grad.abs().max()) > 0, n


# ----------------------------------------------------------------------------------------------------------------------------------
# Whole-network GRADIENT parity at the BASELINE shapes: volo_h12_l18 at full depth and resolution, and the three
# AutoProg stage configurations on the same supernet -- training mode, DropPath masks injected, the mix-token box drawn from a fixed
# seed: loss and EVERY parameter gradient against the fp64 oracle (reference: models/volo.py:644-694, loss/cross_entropy.py:136-156).
def _inject_droppath(model, B, rng):
    """fixed per-sample keep masks for every active Transformer block, in forward order -> {(stage, idx): (m1, m2)} for the oracle"""
    import numpy as np
    masks = {}
    stage = 0
    for net_idx, mod in enumerate(model.network):
        if not hasattr(mod, "__iter__"):
            continue                                   # the Downsample between stage 0 and 1
        for i, blk in enumerate(mod):
            rate = getattr(blk, "drop_prob", 0.0)
            if stage > 0 and rate > 0 and not getattr(blk, "is_identity_layer", False):
                m1 = torch.from_numpy((rng.rand(B) < (1 - rate)).astype(np.float32))
                m2 = torch.from_numpy((rng.rand(B) < (1 - rate)).astype(np.float32))
                masks[(stage, i)] = (m1, m2)
                model.drop_path_rng.queue += [m1, m2]
        stage += 1
    return masks


def _grad_report(model, pref, min_norm=1e-9):
    """-> (all gradients as one vector rel-L2, {name: rel-L2} of every tensor the oracle has a gradient for, names the oracle left without)"""
    num = den = 0.0
    per, none = {}, []
    for n, q in model.named_parameters():
        g = pref[n].grad
        if g is None or float(g.norm()) < min_norm:
            none.append(n)
            assert q.grad is None or float(q.grad.abs().max()) == 0.0, n
            continue
        assert q.grad is not None, n
        d = q.grad.detach().double().cpu() - g
        num += float(d.pow(2).sum()); den += float(g.pow(2).sum())
        per[n] = float(d.norm() / g.norm())
    return (num / den) ** 0.5, per, none


# Measured on MI355X (round 6, gpurun_out/r06d -> profiles/r06_fullnet_gradient_parity.txt), batch 8:
#   (18, 224): outputs 8.5e-3 / 1.2e-2, loss 10.47577 against 10.47500 (7e-5), 251 gradient tensors as one vector 9.8e-3, median 9.0e-3,
#              worst 0.116 / 0.099 (the first two BatchNorm biases), 0.094 - 0.063 (stem convolution weights); everything behind the stem < 6e-2
#   (9, 128) / (12, 160) / (15, 192): one vector 1.04e-2 / 9.0e-3 / 9.5e-3, worst stem tensor 0.101 / 0.101 / 0.108
# Bounds: 6e-2 per tensor behind the stem (the D1-width test's), 0.15 in the stem (1.3 x the worst measured; the 0.12 VERDICT r5 named is met by
# the measurements, the margin is for the pool's boxes), 3e-2 for all gradients as one vector.
FULLNET_TENSOR_TOL, FULLNET_STEM_TOL, FULLNET_GLOBAL_TOL = 6e-2, 0.15, 3e-2


@pytest.mark.parametrize("l,r", [(18, 224), (9, 128), (12, 160), (15, 192)])
def test_d1_train_step_loss_and_every_gradient_vs_oracle(l, r):
    """BASELINE configs[1] (l = 18, r = 224: VOLO-D1 at full depth and resolution) and configs[2]'s stage shapes (9, 128) / (12, 160) /
    (15, 192) on the same volo_h12_l18 supernet: batch 8, training mode (batch-statistics BatchNorm, mix-token, DropPath 0.1 with injected
    masks), token-label loss -- the two outputs, the loss and every parameter gradient against the fp64 oracle; identity layers of a
    sub-network receive no gradient."""
    import numpy as np
    from oracle import ref_cpu as R
    from autoprog_amd.models import create_model
    from autoprog_amd.loss import TokenLabelCrossEntropy
    torch.manual_seed(3)
    model = create_model("model_variant", variant="volo_h12_l18", num_classes=1000, img_size=224, drop_path_rate=0.1).cuda().train()
    skip = None
    if l != 18:
        mask = model.set_sample_config(dict(layer_num=l, min_layer_num=9, max_layer_num=18, input_size=r, token_label_size=r // 16))
        model.set_drop_path_rate(0.1)
        skip = R.skip_layer_table(l, 9, 18)
        assert [sorted(s) for s in mask.skip] == [sorted(s) for s in skip]
    B = 8
    g = torch.Generator().manual_seed(l)
    x = torch.randn(B, 3, r, r, generator=g).cuda()
    n_tok = (r // 16) ** 2
    target = torch.softmax(torch.randn(B, 1000, 2 + n_tok, generator=g) * 3, dim=1).cuda()
    masks = _inject_droppath(model, B, np.random.RandomState(100 + l))
    assert masks and not all(bool(m.all()) for pair in masks.values() for m in pair), "the injected masks drop nothing"
    np.random.seed(l)
    out = model(x)
    loss = TokenLabelCrossEntropy(dense_weight=0.5, cls_weight=1.0, classes=1000)(out, target)
    loss.backward()
    torch.cuda.synchronize()
    assert not model.drop_path_rng.queue, "injected masks left over: the forward consumed fewer DropPath sites than the oracle has"

    p = {k: v.detach().double().cpu() for k, v in model.state_dict().items()}
    for v in p.values():
        if v.dtype.is_floating_point:
            v.requires_grad_(True)
    lam, box = R.draw_mix_box((B, r // 8, r // 8, 192), 2, 1.0, np.random.RandomState(l))
    assert tuple(int(v) for v in out[2]) == tuple(box)
    # (the rates the oracle derives from drop_path_rate are those of the FULL network, models/volo.py:428-437; a sub-network's active blocks are
    # renumbered by set_drop_path_rate -- the oracle only needs keep = 1 - rate per block, so it gets the module's own rates through the masks'
    # keep probabilities: both sides use blk.drop_prob)
    arch = R.variant_arch("volo_h12_l18")
    keeps = {}
    stage = 0
    for mod in model.network:
        if hasattr(mod, "__iter__"):
            for i, blk in enumerate(mod):
                keeps[(stage, i)] = 1.0 - getattr(blk, "drop_prob", 0.0)
            stage += 1
    ref = _oracle_train_forward(R, p, x.double().cpu(), arch, (lam, box), skip, masks, keeps)
    ref_loss = R.token_label_ce(ref, target.double().cpu(), 0.5, 1.0)
    ref_loss.backward()
    e_cls, e_aux = rel(out[0], ref[0].detach()), rel(out[1], ref[1].detach())
    e_loss = abs(float(loss.detach()) - float(ref_loss.detach())) / float(ref_loss.detach())
    glob, per, none = _grad_report(model, p)
    worst = sorted(((round(v, 4), k) for k, v in per.items()), reverse=True)[:6]
    print("(l, r) = (%d, %d): outputs %.2e / %.2e, loss %.5f (oracle %.5f, rel %.1e), %d gradient tensors as one vector %.3e, median %.3e, worst %s"
          % (l, r, e_cls, e_aux, float(loss), float(ref_loss), e_loss, len(per), glob, sorted(per.values())[len(per) // 2], worst))
    assert e_cls < 3e-2 and e_aux < 3e-2, (e_cls, e_aux)
    assert e_loss < 2e-3, e_loss
    assert glob < FULLNET_GLOBAL_TOL, glob
    bad = {k: v for k, v in per.items() if v > (FULLNET_STEM_TOL if k.startswith("patch_embed.conv") else FULLNET_TENSOR_TOL)}
    assert not bad, sorted(bad.items(), key=lambda kv: -kv[1])[:10]
    named = dict(model.named_parameters())
    for n in none:                                   # identity layers of the sub-network (and nothing else)
        assert named[n].grad is None or float(named[n].grad.abs().max()) == 0.0, n
    expect_none = 0 if skip is None else sum(len(s) for s in skip)
    assert len({n.split(".")[1] + "." + n.split(".")[2] for n in none if n.startswith("network.")}) == expect_none, (none[:5], expect_none)


def _oracle_train_forward(R, p, x, arch, mix, skip, masks, keeps, bn_train=None):
    """R.volo_forward in training mode with the MODULE's DropPath rates: the oracle's own rate formula is that of the full network
    (models/volo.py:428-437); for a sub-network set_drop_path_rate renumbers the active blocks, so the rate of block (s, i) is passed in
    through drop_path_rate = 0 and per-block masks scaled by the block's keep probability -- R.transformer takes (masks, keep)."""
    orig = R.transformer

    def transformer_with_module_rates(xx, pp, pre, heads, m, keep, *a, **kw):
        s_net, i = int(pre.split(".")[1]), int(pre.split(".")[2])
        key = (s_net - 1 if s_net >= 2 else s_net, i)
        return orig(xx, pp, pre, heads, masks.get(key), keeps.get(key, 1.0), *a, **kw)
    R.transformer = transformer_with_module_rates
    try:
        return R.volo_forward(p, x, train=True, mix=mix, skip=skip, dp_masks=None, drop_path_rate=0.0, bn_train=bn_train, **arch)
    finally:
        R.transformer = orig


def test_d1_b128_train_step_slice_loss_and_gradients_vs_oracle():
    """The training step at the BENCH size -- volo_h12_l18, 224 px, batch 128, mix-token, DropPath 0.1 -- with the stem's BatchNorm on its
    running statistics (model.train(); model.patch_embed.eval()): the samples are then independent but for the mix-token partner b <-> 127 - b,
    so a slice of four such pairs can be checked against the oracle run on those 8 images alone: the slice's outputs, its token-label loss,
    and -- with the loss taken on the slice only, so that the other 120 images contribute zero -- EVERY parameter gradient of a backward pass
    that runs every kernel at its batch-128 shape."""
    import numpy as np
    from oracle import ref_cpu as R
    from autoprog_amd.models import create_model
    from autoprog_amd.loss import TokenLabelCrossEntropy
    torch.manual_seed(11)
    B = 128
    model = create_model("model_variant", variant="volo_h12_l18", num_classes=1000, img_size=224, drop_path_rate=0.1).cuda().train()
    with torch.no_grad():
        for m in model.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.normal_(0, 0.1); m.running_var.uniform_(0.8, 1.2)
    model.patch_embed.eval()
    # (the HIP stem's backward exists for batch statistics only: PatchEmbed -- its three convolutions, their BatchNorms and the projection that
    # applies the last of them -- takes no gradient here; it has the batch-8 tests above.  Everything behind it does: 26.4 of the 26.6 M parameters)
    frozen = [n for n, q in model.named_parameters() if n.startswith("patch_embed.")]
    for n, q in model.named_parameters():
        if n in frozen:
            q.requires_grad_(False)
    g = torch.Generator().manual_seed(2)
    x = torch.randn(B, 3, 224, 224, generator=g).cuda()
    half = [0, 17, 33, 63]
    idx = half + [B - 1 - i for i in reversed(half)]            # flipping the slice = flipping the batch, restricted to the slice
    target = torch.softmax(torch.randn(8, 1000, 198, generator=g) * 3, dim=1).cuda()
    masks = _inject_droppath(model, B, np.random.RandomState(7))
    np.random.seed(4)
    x_cls, x_aux, box = model(x)
    sel = torch.tensor(idx, device="cuda")
    loss = TokenLabelCrossEntropy(dense_weight=0.5, cls_weight=1.0, classes=1000)((x_cls[sel], x_aux[sel], box), target)
    loss.backward()
    torch.cuda.synchronize()

    p = {k: v.detach().double().cpu() for k, v in model.state_dict().items()}
    for k, v in p.items():
        if v.dtype.is_floating_point and "running_" not in k and k not in frozen:
            v.requires_grad_(True)
    lam, rbox = R.draw_mix_box((B, 28, 28, 192), 2, 1.0, np.random.RandomState(4))
    assert tuple(int(v) for v in box) == tuple(rbox)
    keeps = {}
    stage = 0
    for mod in model.network:
        if hasattr(mod, "__iter__"):
            for i, blk in enumerate(mod):
                keeps[(stage, i)] = 1.0 - getattr(blk, "drop_prob", 0.0)
            stage += 1
    smasks = {k: (m1[idx], m2[idx]) for k, (m1, m2) in masks.items()}
    ref = _oracle_train_forward(R, p, x[sel].double().cpu(), R.variant_arch("volo_h12_l18"), (lam, rbox), None, smasks, keeps, bn_train=False)
    ref_loss = R.token_label_ce(ref, target.double().cpu(), 0.5, 1.0)
    ref_loss.backward()
    e_cls, e_aux = rel(x_cls[sel], ref[0].detach()), rel(x_aux[sel], ref[1].detach())
    e_loss = abs(float(loss.detach()) - float(ref_loss.detach())) / float(ref_loss.detach())
    glob, per, none = _grad_report(model, p)
    assert sorted(none) == sorted(frozen), (none, frozen)
    worst = sorted(((round(v, 4), k) for k, v in per.items()), reverse=True)[:6]
    print("B = 128 slice: outputs %.2e / %.2e, loss %.5f (oracle %.5f, rel %.1e), %d gradient tensors as one vector %.3e, median %.3e, worst %s"
          % (e_cls, e_aux, float(loss), float(ref_loss), e_loss, len(per), glob, sorted(per.values())[len(per) // 2], worst))
    assert e_cls < 3e-2 and e_aux < 3e-2, (e_cls, e_aux)
    assert e_loss < 2e-3, e_loss
    assert glob < FULLNET_GLOBAL_TOL, glob
    bad = {k: v for k, v in per.items() if v > (FULLNET_STEM_TOL if k.startswith("patch_embed.conv") else FULLNET_TENSOR_TOL)}
    assert not bad, sorted(bad.items(), key=lambda kv: -kv[1])[:10]
