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
    a, b = a.double(), b.double()
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
