"""Kernel-level parity: every C-ABI entry point (through autoprog_amd.ops -> ctypes ->
libautoprog_hip.so) against the CPU oracle on the same seeded, bf16-rounded inputs.

Tolerances: bf16 outputs <= 1e-2 rel-L2 per tensor against
the fp32/fp64 oracle evaluated on the bf16-rounded inputs (expected ~3e-3 = bf16 rounding of
the output); fp32-accumulated weight gradients <= 3e-3; integer/bookkeeping ops exact.
"""
import math
import os
import sys

import pytest
import torch

from oracle import ref_cpu as R

pytestmark = pytest.mark.gpu
TOL_BF16 = 1e-2
TOL_F32 = 3e-3


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "GPU tests need a real MI355X"
    from autoprog_amd import ops as _ops
    return _ops


def rnd(*shape, scale=1.0, seed=0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(torch.bfloat16)


def rel(a, b):
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def dev(t):
    return t.cuda().contiguous()


# ------------------------------------------------------------------------------------------
def test_abi_loaded(ops):
    from autoprog_amd._lib import lib, LIB_PATH
    assert lib.ap_abi_version() == 7
    assert LIB_PATH.endswith("libautoprog_hip.so")


def test_casts(ops):
    x = torch.randn(1000, 37)
    y = ops.cast_bf16(dev(x))
    assert torch.equal(y.cpu(), x.to(torch.bfloat16))
    assert torch.equal(ops.cast_f32(y).cpu(), x.to(torch.bfloat16).float())
    w = torch.randn(486, 192)
    wt = ops.cast_transpose_bf16(dev(w))
    assert wt.shape == (192, 488)
    assert torch.equal(wt[:, :486].cpu(), w.t().to(torch.bfloat16))
    assert float(wt[:, 486:].abs().sum()) == 0.0


@pytest.mark.parametrize("rows,C,eps", [(1000, 192, 1e-5), (777, 384, 1e-5), (50, 32, 1e-5), (33, 64, 1e-6), (64, 768, 1e-6),
                                       (20, 1152, 1e-5)])
def test_layernorm(ops, rows, C, eps):
    x = rnd(rows, C, scale=2.0, seed=1) + 0.5
    g = torch.randn(C, generator=torch.Generator().manual_seed(2)) * 0.3 + 1
    b = torch.randn(C, generator=torch.Generator().manual_seed(3)) * 0.3
    dy = rnd(rows, C, seed=4)
    dres = rnd(rows, C, seed=5)
    xr = x.double().requires_grad_(True)
    gr, br = g.double().requires_grad_(True), b.double().requires_grad_(True)
    yr = R.layernorm(xr, gr, br, eps)
    yr.backward(dy.double())
    y, mean, rstd = ops.layernorm_fwd(dev(x), dev(g), dev(b), eps)
    assert rel(y, yr) < TOL_BF16
    dg = torch.zeros(C, device="cuda")
    db = torch.zeros(C, device="cuda")
    dx = ops.layernorm_bwd(dev(dy), dev(x), dev(g), mean, rstd, dev(dres), dg, db)
    assert rel(dx, xr.grad + dres.double()) < TOL_BF16
    assert rel(dg, gr.grad) < TOL_F32 and rel(db, br.grad) < TOL_F32
    dx2 = ops.layernorm_bwd(dev(dy), dev(x), dev(g), mean, rstd, None, dg, db)
    assert rel(dx2, xr.grad) < TOL_BF16


@pytest.mark.parametrize("M,N,K", [(256, 128, 64), (1000, 192, 192), (300, 486, 192), (128, 1152, 384), (77, 1000, 384),
                                   (512, 384, 1152), (130, 32, 32), (64, 96, 96), (200, 16, 64), (392, 384, 768),
                                   (2304, 192, 192), (4100, 384, 384), (2500, 1152, 384), (3000, 576, 192), (70000, 192, 576),
                                   (128, 384, 1000), (100, 64, 40), (256, 392, 1064)])       # few rows, K not a multiple of 32 (the head's input gradient)
def test_gemm_nt_plain_and_bias(ops, M, N, K):
    a, w = rnd(M, K, seed=1), rnd(N, K, scale=K ** -0.5, seed=2)
    bias = torch.randn(N, generator=torch.Generator().manual_seed(3))
    ref = a.double() @ w.double().t()
    out = ops.gemm_nt(dev(a), dev(w))
    assert out.shape == (M, ops.round_up(N, 8))
    assert rel(out[:, :N], ref) < TOL_BF16
    out = ops.gemm_nt(dev(a), dev(w), bias=dev(bias))
    assert rel(out[:, :N], ref + bias.double()) < TOL_BF16


@pytest.mark.parametrize("M,N,K", [(392, 576, 192), (128, 1152, 384), (1, 384, 384), (200, 200, 1152), (256, 32, 32),
                                   (4352, 576, 192), (4200, 1152, 384), (4100, 384, 1152)])
def test_gemm_nt_epilogues(ops, M, N, K):
    """second to fifth shapes: the few-rows kernel (M <= 256: class-attention blocks, cls heads); the last three: the persistent
    8-phase kernel (M >= 4096; 256 x 192 and 256 x 256 tiles, ragged last row tile)"""
    rps = max(1, M // 2)
    a, w = rnd(M, K, seed=1), rnd(N, K, scale=K ** -0.5, seed=2)
    bias = torch.randn(N, generator=torch.Generator().manual_seed(3))
    res = rnd(M, N, seed=4)
    rs = torch.tensor([0.0, 1.0 / 0.9, 1.0 / 0.9])
    lin = a.double() @ w.double().t() + bias.double()
    # gelu with pre-activation side output
    assert N % 8 == 0
    h = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    out = ops.gemm_nt(dev(a), dev(w), bias=dev(bias), gelu=True, preact_out=h)
    assert rel(h, lin) < TOL_BF16
    assert rel(out, R.gelu(h.double().cpu())) < TOL_BF16          # activation of the ROUNDED pre-activation
    # residual + row scale (DropPath)
    out = ops.gemm_nt(dev(a), dev(w), bias=dev(bias), row_scale=dev(rs), rows_per_scale=rps, residual=dev(res))
    ref = lin * rs.double().repeat_interleave(rps)[:M, None] + res.double()
    assert rel(out, ref) < TOL_BF16
    assert torch.equal(out[:rps].cpu(), res[:rps])               # dropped samples pass the residual through exactly
    # dgelu epilogue
    hh = rnd(M, N, seed=6)
    hr = hh.double().requires_grad_(True)
    R.gelu(hr).backward(torch.ones(M, N, dtype=torch.float64))
    out = ops.gemm_nt(dev(a), dev(w), dgelu_of=dev(hh))
    assert rel(out, (a.double() @ w.double().t()) * hr.grad) < TOL_BF16
    # the same pair with the activation DERIVATIVE stored by the forward (gelu = 2) and multiplied in by the backward (mul_by): what the
    # blocks use -- gelu'(h) is all the backward needs of h (autograd of models/volo.py:157)
    gp = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    out2 = ops.gemm_nt(dev(a), dev(w), bias=dev(bias), gelu=True, preact_out=gp, preact_grad=True, row_scale=dev(rs), rows_per_scale=rps)
    hb = lin.to(torch.bfloat16).double().requires_grad_(True)             # the rounded pre-activation the activation is applied to
    R.gelu(hb).backward(torch.ones(M, N, dtype=torch.float64))
    assert rel(gp, hb.grad) < TOL_BF16
    assert rel(out2, R.gelu(hb.detach()) * rs.double().repeat_interleave(rps)[:M, None]) < TOL_BF16
    out3 = ops.gemm_nt(dev(a), dev(w), mul_by=gp, row_scale=dev(rs), rows_per_scale=rps)
    assert rel(out3, (a.double() @ w.double().t()) * gp.double().cpu() * rs.double().repeat_interleave(rps)[:M, None]) < TOL_BF16
    # round 5 (ABI 6): the derivative as 8-bit fixed-point codes (gelu = 3 / mul_by8): code = clamp(rint(202 g') + 26, 0, 255) -- EXACT against
    # that formula on the bf16-rounded pre-activation except where fp32 rounding puts 202 g' + 26 within 1e-3 of a half-integer, the
    # output identical to the gelu = 2 launch, and the backward multiplies by (code - 26) / 202
    codes = torch.empty(M, N, dtype=torch.uint8, device="cuda")
    out4 = ops.gemm_nt(dev(a), dev(w), bias=dev(bias), gelu=True, preact_out=codes, preact_grad=2, row_scale=dev(rs), rows_per_scale=rps)
    # (on the 8-phase kernel the gelu = 3 row phase takes Phi(h) from the 4096-entry table in LDS, kept to 24 bits: a product h * Phi(h)
    # that sits within 2^-16 of a bf16 rounding tie may fall the other way -- at most one bf16 step, in at most 1 % of the elements)
    if not torch.equal(out4, out2):
        d4 = (out4.float() - out2.float()).abs()
        assert float((d4 > 0).float().mean()) < 1e-2 and bool((d4 <= out2.float().abs() * 2.0 ** -7 + 1e-30).all())
    t = hb.grad * ops.GELU_CODE_SCALE + ops.GELU_CODE_ZERO
    want = t.round().clamp(0, 255)
    got = codes.cpu().double()
    off = got != want
    # (the kernel rounds ITS fp32 pre-activation to bf16; the fp64 one of this test can fall on the other side of a tie for an element or
    # two: there gelu' differs by up to one bf16 step of h -- below one code)
    assert int(off.sum()) <= 1e-3 * M * N and float((got - t).abs().max()) < 1.0, (int(off.sum()), float((got - t).abs().max()))
    dec = (got - ops.GELU_CODE_ZERO) / ops.GELU_CODE_SCALE
    assert float((dec - hb.grad).abs().max()) <= 1.0 / ops.GELU_CODE_SCALE
    near = ((got - t).abs() <= 0.5 + 2e-3).double().mean()
    assert float(near) >= 1.0 - 1e-3, float(near)
    # against the launch's OWN derivative (the gelu = 2 output, a bf16 rounding of the same fp32 value): within half a code + that rounding
    assert float((got - (gp.double().cpu() * ops.GELU_CODE_SCALE + ops.GELU_CODE_ZERO)).abs().max()) <= 0.5 + ops.GELU_CODE_SCALE * 2.0 ** -8
    out5 = ops.gemm_nt(dev(a), dev(w), mul_by=codes, row_scale=dev(rs), rows_per_scale=rps)
    assert rel(out5, (a.double() @ w.double().t()) * dec * rs.double().repeat_interleave(rps)[:M, None]) < TOL_BF16
    with pytest.raises(Exception):
        ops.gemm_nt(dev(a), dev(w), bias=dev(bias), gelu=True, preact_out=gp, preact_grad=2)          # codes need a uint8 tensor


@pytest.mark.parametrize("B,H,W,C", [(3, 8, 8, 192), (2, 7, 7, 64), (5, 9, 6, 384), (16, 28, 28, 192)])
def test_layernorm_bwd_with_the_average_pools_gradient(ops, B, H, W, C):
    """ap_layernorm_bwd_partial_pool (round 5): the backward of the 2 x 2 ceil-mode average pool that reads the same LayerNorm output
    (OutlookAttention, models/volo.py:75,87) applied to the incoming gradient INSIDE the LayerNorm backward kernel, against fp64: the pooled
    gradient spread over the cells' pixels (divided by the CLIPPED count on odd grids), added to dy, LayerNorm backward, + residual
    gradient; dgamma / dbeta through the deferred reduction.  And against the two-launch form it replaces (ap_avgpool2_bwd_acc rounds the
    sum to bf16 first: equal within that rounding)."""
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(B * 100 + H)
    x = torch.randn(B, H, W, C, generator=g).bfloat16()
    dy = torch.randn(B, H, W, C, generator=g).bfloat16()
    dres = torch.randn(B, H, W, C, generator=g).bfloat16()
    h, w = (H + 1) // 2, (W + 1) // 2
    dp = torch.randn(B, h, w, C, generator=g).bfloat16()
    gamma = 1 + 0.2 * torch.randn(C, generator=g)
    beta = 0.1 * torch.randn(C, generator=g)
    x64 = x.double().requires_grad_(True)
    g64, b64 = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    y = F.layer_norm(x64, (C,), g64, b64, 1e-5)
    pooled = F.avg_pool2d(y.permute(0, 3, 1, 2), 2, 2, ceil_mode=True, count_include_pad=False).permute(0, 2, 3, 1)
    ((y * dy.double()).sum() + (pooled * dp.double()).sum()).backward()
    want_dx = x64.grad + dres.double()
    xd = dev(x)
    yk, mean, rstd = ops.layernorm_fwd(xd.view(-1, C), dev(gamma), dev(beta), 1e-5)
    dg, db = torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
    items = []
    dx = ops.layernorm_bwd(dev(dy).view(-1, C), xd.view(-1, C), dev(gamma), mean, rstd, dev(dres).view(-1, C), dg, db, defer=items, pool=(dev(dp), (B, H, W)))
    if os.environ.get("AP_LN_BWD_PF", "1") == "0":
        assert dx is None                      # the pool's gradient rides in the pipelined kernel only: the caller falls back to two launches
        return
    assert dx is not None
    ops.layernorm_bwd_reduce_batched(items)
    assert rel(dx.view(B, H, W, C), want_dx) < TOL_BF16
    assert rel(dg, g64.grad) < 1e-4 and rel(db, b64.grad) < 1e-4
    # the two launches it replaces
    dy2 = dev(dy).clone()
    ops.avgpool2_bwd_acc(dev(dp), dy2)
    dg2, db2 = torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
    dx2 = ops.layernorm_bwd(dy2.view(-1, C), xd.view(-1, C), dev(gamma), mean, rstd, dev(dres).view(-1, C), dg2, db2)
    assert rel(dx, dx2.cpu()) < 6e-3 and rel(dg, dg2.cpu()) < 6e-3


@pytest.mark.parametrize("M,K", [(25088, 384), (25088, 1152), (24999, 384), (18432, 1152), (4100, 384)])
def test_gemm_nt_224_row_tiles(ops, M, K):
    """round 5: the 224-row instantiation of the 8-phase kernel (csrc/gemm8p.h BM = 224), taken where it puts more CUs to work inside one
    round of the persistent grid -- the N = 384 products of a VOLO-D1 transformer block at 25088 rows (196 -> 224 tiles): every flavour
    it is instantiated for (plain, row scale, bias + residual, bias + row scale + residual) against fp64, with a ragged last row tile, and
    bit-equal to the 256-row tiles (AP_GEMM_BM224=0 is checked by the environment-switch test): the accumulation order per element is the same."""
    N = 384
    rps = 196
    a, w = rnd(M, K, seed=1), rnd(N, K, scale=K ** -0.5, seed=2)
    bias = torch.randn(N, generator=torch.Generator().manual_seed(3))
    res = rnd(M, N, seed=4)
    rs = (torch.rand((M + rps - 1) // rps, generator=torch.Generator().manual_seed(5)) > 0.2).float() / 0.8
    lin = a.double() @ w.double().t()
    rsr = rs.double().repeat_interleave(rps)[:M, None]
    da, dw_ = dev(a), dev(w)
    assert rel(ops.gemm_nt(da, dw_), lin) < TOL_BF16
    assert rel(ops.gemm_nt(da, dw_, row_scale=dev(rs), rows_per_scale=rps), lin * rsr) < TOL_BF16
    assert rel(ops.gemm_nt(da, dw_, bias=dev(bias), residual=dev(res)), lin + bias.double() + res.double()) < TOL_BF16
    out = ops.gemm_nt(da, dw_, bias=dev(bias), row_scale=dev(rs), rows_per_scale=rps, residual=dev(res))
    assert rel(out, (lin + bias.double()) * rsr + res.double()) < TOL_BF16
    dropped = (rs == 0).repeat_interleave(rps)[:M]
    assert torch.equal(out.cpu()[dropped], res[dropped])            # dropped samples pass the residual through exactly


@pytest.mark.parametrize("M,N", [(16384, 576), (25088 * 4, 576), (20480, 192), (16448, 384)])
def test_gemm_nt_weight_stationary_k192(ops, M, N):
    """round 5: csrc/gemm_ws.h, the weight-stationary kernel for K = 192 (the Outlooker's MLP, models/volo.py:156-163 at the 192-channel
    stage: fc1 + GELU with the 8-bit derivative codes, and the input gradient of fc2 times those codes), taken from 16384 rows on.  Against
    fp64 (outputs TOL_BF16, codes within one step of the fp64 derivative) -- and BIT-EQUAL to the 8-phase kernel, which the same problem
    cut into 8192-row pieces still runs on: same K order, same rounding points."""
    import torch.nn.functional as F
    K = 192
    a, w = rnd(M, K, seed=1), rnd(N, K, scale=K ** -0.5, seed=2)
    bias = torch.randn(N, generator=torch.Generator().manual_seed(3)) * 0.3
    da, dw_, db = dev(a), dev(w), dev(bias)
    codes = torch.empty(M, N, device="cuda", dtype=torch.uint8)
    y = ops.gemm_nt(da, dw_, bias=db, gelu=True, preact_out=codes, preact_grad=2)
    h = (a.double() @ w.double().t() + bias.double()).to(torch.bfloat16).double()
    assert rel(y, F.gelu(h)) < TOL_BF16
    hg = h.clone().requires_grad_(True)
    F.gelu(hg).sum().backward()
    want = torch.clamp(torch.round(hg.grad * ops.GELU_CODE_SCALE) + ops.GELU_CODE_ZERO, 0, 255)
    # the kernel rounds its own fp32 accumulation of h to bf16: where that lands on the other side of a rounding tie the fp64 code differs by more
    dcode = (codes.cpu().double() - want).abs()
    assert float((dcode <= 1).double().mean()) > 0.999 and float(dcode.max()) < 6
    mul = torch.randint(0, 256, (M, N), dtype=torch.uint8, generator=torch.Generator().manual_seed(4))
    dmul = dev(mul)
    d = ops.gemm_nt(da, dw_, mul_by=dmul)
    assert rel(d, (a.double() @ w.double().t()) * ((mul.double() - ops.GELU_CODE_ZERO) / ops.GELU_CODE_SCALE)) < TOL_BF16
    # the same problem in pieces below the kernel's row threshold (-> the 8-phase kernel, unless an environment-switch run has taken it away)
    if os.environ.get("AP_GEMM_8P", "1") == "0":
        return
    ys, cs, ds = [], [], []
    cuts = list(range(0, M, 8192)) + [M]
    if cuts[-1] - cuts[-2] < 4096:            # (a last piece of fewer than 4096 rows would leave the 8-phase kernel too: it joins the one before)
        del cuts[-2]
    for m0, m1 in zip(cuts[:-1], cuts[1:]):
        c = torch.empty(m1 - m0, N, device="cuda", dtype=torch.uint8)
        ys.append(ops.gemm_nt(da[m0:m1].contiguous(), dw_, bias=db, gelu=True, preact_out=c, preact_grad=2))
        cs.append(c)
        ds.append(ops.gemm_nt(da[m0:m1].contiguous(), dw_, mul_by=dmul[m0:m1].contiguous()))
    assert torch.equal(y, torch.cat(ys)) and torch.equal(codes, torch.cat(cs)) and torch.equal(d, torch.cat(ds))


def test_eight_bit_gelu_derivative_against_the_exact_derivative(ops):
    """the stored gelu' lives on an 8-bit grid (step 1/202) and the rounding-matched oracle follows it, so THIS test keeps the grid's own
    error visible: the input gradient of an MLP's fc2 through the stored codes (gelu = 3 forward, mul_by8 backward) against fp64 with the EXACT
    derivative of the same bf16 pre-activation.  Element-wise the code is off by <= 1/404 absolute -- relative errors of order one where |gelu'| is
    a few thousandths -- as one tensor the gradient is within 4e-3 of the exact one (measured 2.4e-3; the bf16 derivative of mode 1: 1.6e-3)."""
    import torch.nn.functional as F
    M, C, H = 8192, 384, 1152
    x, w1, w2t = rnd(M, C, seed=1), rnd(H, C, scale=C ** -0.5, seed=2), rnd(H, C, scale=H ** -0.5, seed=3)
    b1 = torch.randn(H, generator=torch.Generator().manual_seed(4)) * 0.3
    dy = rnd(M, C, seed=5)
    codes = torch.empty(M, H, device="cuda", dtype=torch.uint8)
    ops.gemm_nt(dev(x), dev(w1), bias=dev(b1), gelu=True, preact_out=codes, preact_grad=2)
    dh = ops.gemm_nt(dev(dy), dev(w2t), mul_by=codes)
    h = (x.double() @ w1.double().t() + b1.double()).to(torch.bfloat16).double().requires_grad_(True)
    F.gelu(h).sum().backward()
    ref = (dy.double() @ w2t.double().t()) * h.grad
    e = rel(dh, ref)
    dec = (codes.cpu().double() - ops.GELU_CODE_ZERO) / ops.GELU_CODE_SCALE
    worst_abs = float((dec - h.grad).abs().max())
    small = h.grad.abs() < 5e-3
    print("8-bit gelu' against the exact derivative: dL/dh rel-L2 %.2e; code error max %.2e absolute; where |gelu'| < 5e-3 (%.1f %% of the elements) "
          "median relative error %.2f" % (e, worst_abs, 100 * float(small.double().mean()),
                                          float(((dec - h.grad).abs() / h.grad.abs().clamp_min(1e-12))[small].median())))
    # (a code is <= 1/404 from the derivative of the kernel's OWN bf16 h; where the kernel's fp32 sum and the fp64 one round h to different bf16
    # neighbours the derivative itself moves by up to ~3e-3 on top)
    assert worst_abs <= 1.0 / 404 + 4e-3, worst_abs
    assert e < 4e-3, e


@pytest.mark.parametrize("M,drop", [(128, True), (1152, False), (6272, False), (6272, True), (25088, True)])
def test_mlp_fused_is_the_two_launches_bit_for_bit(ops, M, drop):
    """round 6: csrc/mlp_fused.hip -- fc1 -> GELU -> fc2 (+ DropPath scale + residual) of a transformer block (models/volo.py:147-167, :233) in ONE
    launch, and the two input-gradient products of its backward pass in one launch.  Same K order and rounding points as the two ap_gemm_nt
    launches each replaces: hidden activation, gelu' codes, dL/dh and both outputs BIT-IDENTICAL to them (with and without DropPath factors), and
    within bf16 tolerance of fp64.  With ln=: the LayerNorm in front of fc1 inside the same launch -- normalised rows, mean, rstd and everything
    behind them bit-identical to ap_layernorm_fwd followed by the launch above."""
    import torch.nn.functional as F
    if os.environ.get("AP_GEMM_8P", "1") == "0" or os.environ.get("AP_GELU_TABLE", "1") == "0":
        pytest.skip("bit equality is against the 8-phase kernel's table path")
    C, H, N = 384, 1152, (196 if M >= 196 else 64)
    # bit equality is against the 8-phase kernel; launches of fewer than 4096 rows go to other ap_gemm_nt kernels (another summation order, the GELU
    # evaluated instead of looked up): the fused launch of a single block / of nine blocks is checked against fp64 alone
    bit = M >= 4096
    x, w1, w2 = rnd(M, C, seed=1), rnd(H, C, scale=C ** -0.5, seed=2), rnd(C, H, scale=H ** -0.5, seed=3)
    g = torch.Generator().manual_seed(4)
    b1, b2 = torch.randn(H, generator=g) * 0.3, torch.randn(C, generator=g) * 0.3
    res, dy = rnd(M, C, seed=5), rnd(M, C, seed=6)
    keep = (torch.rand((M + N - 1) // N, generator=g) < 0.8).float()
    k2, rs2 = (dev(keep), dev(keep / 0.8)) if drop else (None, None)
    dx_, dw1, dw2, db1, db2, dres, ddy = dev(x), dev(w1), dev(w2), dev(b1), dev(b2), dev(res), dev(dy)
    # forward
    codes0 = torch.empty(M, H, device="cuda", dtype=torch.uint8)
    a0 = ops.gemm_nt(dx_, dw1, bias=db1, gelu=True, preact_out=codes0, preact_grad=2, row_scale=k2, rows_per_scale=N)
    y0 = ops.gemm_nt(a0, dw2, bias=db2, row_scale=rs2, rows_per_scale=N, residual=dres)
    got = ops.mlp_fused(dx_, dw1, dw2, bias1=db1, bias2=db2, row_scale_hidden=k2, row_scale_out=rs2, rows_per_scale=N, residual=dres)
    assert got is not None, "ap_mlp_fused refused a launch it is built for"
    y1, a1, codes1 = got
    assert not bit or (torch.equal(a0, a1) and torch.equal(codes0, codes1) and torch.equal(y0, y1))
    if not bit:
        codes0 = codes1                          # (the backward below reads the fused forward's own codes)
    rows = torch.arange(0, M, 37)
    h = (x[rows].double() @ w1.double().t() + b1.double()).to(torch.bfloat16).double()
    kk = keep[rows // N].double()[:, None] if drop else 1.0
    aref = F.gelu(h) * kk
    assert rel(a1[rows.cuda()], aref) < TOL_BF16
    yref = (aref.to(torch.bfloat16).double() @ w2.double().t() + b2.double()) * (kk / 0.8 if drop else 1.0) + res[rows].double()
    assert rel(y1[rows.cuda()], yref) < TOL_BF16
    # backward: dL/dh = (dy W2) gelu'(code) rs, dL/dx = dL/dh W1
    w2t, w1t = dev(w2.t().contiguous()), dev(w1.t().contiguous())
    dh0 = ops.gemm_nt(ddy, w2t, mul_by=codes0, row_scale=rs2, rows_per_scale=N)
    dx0 = ops.gemm_nt(dh0, w1t)
    dx1, dh1, _ = ops.mlp_fused(ddy, w2t, w1t, backward=True, codes=codes0, row_scale_hidden=rs2, rows_per_scale=N)
    assert not bit or (torch.equal(dh0, dh1) and torch.equal(dx0, dx1))
    gp = (codes0[rows.cuda()].cpu().double() - ops.GELU_CODE_ZERO) / ops.GELU_CODE_SCALE
    dhref = (dy[rows].double() @ w2.double()) * gp * (kk / 0.8 if drop else 1.0)
    assert rel(dh1[rows.cuda()], dhref) < TOL_BF16
    assert rel(dx1[rows.cuda()], dhref.to(torch.bfloat16).double() @ w1.double()) < TOL_BF16
    # the LayerNorm in front of fc1 inside the launch (rows with a mean and a spread of their own; the residual is the LayerNorm's input, as in the block)
    xin = dev((x.float() * (1.0 + torch.rand(M, 1, generator=g) * 3.0) + torch.randn(M, 1, generator=g) * 2.0).to(torch.bfloat16))
    lg, lb = dev(1.0 + 0.3 * torch.randn(C, generator=g)), dev(0.2 * torch.randn(C, generator=g))
    xn0, m0, r0 = ops.layernorm_fwd(xin, lg, lb, 1e-5)
    y0, a0, c0 = ops.mlp_fused(xn0, dw1, dw2, bias1=db1, bias2=db2, row_scale_hidden=k2, row_scale_out=rs2, rows_per_scale=N, residual=xin)
    got = ops.mlp_fused(None, dw1, dw2, bias1=db1, bias2=db2, row_scale_hidden=k2, row_scale_out=rs2, rows_per_scale=N, residual=xin, ln=(xin, lg, lb, 1e-5))
    if os.environ.get("AP_MLP_FUSED_V") == "1":
        assert got is None                      # the first structure has no LayerNorm: refused, the caller launches ap_layernorm_fwd
        return
    assert got is not None
    y1, a1, c1, xn1, m1, r1 = got
    assert torch.equal(xn0, xn1) and torch.equal(m0, m1) and torch.equal(r0, r1)
    assert torch.equal(a0, a1) and torch.equal(c0, c1) and torch.equal(y0, y1)
    xr = xin[rows.cuda()].cpu().double()
    lnref = (xr - xr.mean(1, keepdim=True)) / (xr.var(1, unbiased=False, keepdim=True) + 1e-5).sqrt() * lg.cpu().double() + lb.cpu().double()
    assert rel(xn1[rows.cuda()], lnref) < TOL_BF16
    # launches outside what the kernel is built for are refused, not mis-computed
    assert ops.mlp_fused(dev(rnd(1000, C, seed=9)), dw1, dw2) is None


@pytest.mark.parametrize("M,N1,N2", [(1024, 128, 128), (1000, 192, 576), (3000, 486, 192), (25088 // 8, 1152, 384), (130, 1000, 384),
                                     (77, 32, 96), (500, 16, 64)])
def test_gemm_tn_acc(ops, M, N1, N2):
    l1, l2 = ops.round_up(N1, 8), ops.round_up(N2, 8)
    a, b = rnd(M, l1, seed=1), rnd(M, l2, seed=2)
    c0 = torch.randn(N1, N2, generator=torch.Generator().manual_seed(3))
    c = dev(c0)
    cs0 = torch.randn(N1, generator=torch.Generator().manual_seed(4))
    cs = dev(cs0)
    ops.gemm_tn_acc(dev(a), dev(b), c, colsum=cs)
    ref = c0.double() + a[:, :N1].double().t() @ b[:, :N2].double()
    assert rel(c, ref) < TOL_F32
    assert rel(cs, cs0.double() + a[:, :N1].double().sum(0)) < TOL_F32
    c2 = dev(c0)
    ops.gemm_tn_acc(dev(a), dev(b), c2)                      # without the fused bias gradient
    assert rel(c2, ref) < TOL_F32


def test_gemm_tn_acc_grouped(ops):
    """several weight gradients in one launch == the same problems one by one (shapes of a VOLO block, ragged sizes,
    different token counts, with and without the fused bias gradient; more problems than one launch holds)"""
    shapes = [(3136, 1152, 384, True), (3136, 384, 384, True), (3136, 1152, 384, False), (3136, 384, 1152, True),
              (1000, 486, 192, True), (77, 32, 96, False), (4000, 192, 192, True), (130, 1000, 384, True), (500, 16, 64, True),
              (2048, 576, 192, False)]
    probs, refs = [], []
    for i, (M, N1, N2, with_cs) in enumerate(shapes):
        l1, l2 = ops.round_up(N1, 8), ops.round_up(N2, 8)
        a, b = rnd(M, l1, seed=10 + i), rnd(M, l2, seed=40 + i)
        c0 = torch.randn(N1, N2, generator=torch.Generator().manual_seed(70 + i))
        cs0 = torch.randn(N1, generator=torch.Generator().manual_seed(90 + i))
        c, cs = dev(c0), (dev(cs0) if with_cs else None)
        probs.append((dev(a), dev(b), c, N1, N2, cs))
        refs.append((c0.double() + a[:, :N1].double().t() @ b[:, :N2].double(), cs0.double() + a[:, :N1].double().sum(0)))
    ops.gemm_tn_acc_grouped(probs)
    for (a, b, c, n1, n2, cs), (rc, rcs) in zip(probs, refs):
        assert rel(c, rc) < TOL_F32
        if cs is not None:
            assert rel(cs, rcs) < TOL_F32
    with pytest.raises(Exception):
        ops.gemm_tn_acc_grouped([(probs[0][0], probs[1][1][:100], probs[0][2], None, None, None)])      # token counts differ


def test_gemm_tn_grouped_deterministic_and_weighted_colsum(ops):
    """AP_DETERMINISTIC path (stored partial tiles + ordered reduce): bit-identical across runs and equal to the reference;
    the atomic path agrees to fp32 rounding.  Also the mask-weighted column sum (DropPath bias gradient): colsum += s * sum_m w[m] A[m,n], and the product factor alpha."""
    M, N1, N2 = 25088 // 4 + 3, 384, 1152                # ragged token count: the last 16-byte chunk of the weights is padded
    a, b = dev(rnd(M, N1, seed=1)), dev(rnd(M, N2, seed=2))
    w = (torch.rand(M, generator=torch.Generator().manual_seed(3)) < 0.8).to(torch.bfloat16)
    wpad = torch.cat([w, torch.zeros((-M) % 8, dtype=torch.bfloat16)])
    ref_c = 0.8 * (a.double().cpu().t() @ b.double().cpu())
    ref_cs = 1.25 * (w.double()[:, None] * a.double().cpu()).sum(0)
    outs = []
    old = ops.deterministic
    try:
        for det in (True, True, False):
            ops.deterministic = det
            c = torch.zeros(N1, N2, device="cuda")
            cs = torch.zeros(N1, device="cuda")
            ops.gemm_tn_acc_grouped([(a, b, c, N1, N2, cs, dev(wpad), 1.25, 0.8), (b, a, torch.zeros(N2, N1, device="cuda"), N2, N1, None)])
            assert rel(c, ref_c) < TOL_F32 and rel(cs, ref_cs) < TOL_F32
            outs.append((c.clone(), cs.clone()))
    finally:
        ops.deterministic = old
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])          # deterministic: bitwise
    assert rel(outs[2][0], outs[0][0]) < 1e-5


def test_gemm_tn_8p_tile_kernel(ops):
    """the 192 x 192-tile LDS-DMA weight-gradient kernel (csrc/gemm_tn8p.h: widths that are multiples of 192, whole 64-token K-tiles,
    M >= 4096) against fp64: a block-shaped group with different token counts, accumulation into non-zero C, the product factor,
    plain and mask-weighted column sums, operands with padded leading dimensions; and against the 128 x 128-tile kernel on the
    same problems (AP_GEMM_TN_8P=0 is re-run by the environment-switch test)"""
    specs = [(6272, 384, 1152, "w", 0.8), (6272, 1152, 384, "1", 1.0), (4096, 384, 384, "w", 1.0), (8192, 192, 576, None, 1.0),
             (4160, 576, 192, "1", 0.5)]
    probs, refs = [], []
    for i, (M, N1, N2, cs_kind, alpha) in enumerate(specs):
        la, lb = N1 + (8 if i % 2 else 0), N2 + (16 if i == 3 else 0)
        a, b = rnd(M, la, seed=20 + i), rnd(M, lb, seed=50 + i)
        c0 = torch.randn(N1, N2, generator=torch.Generator().manual_seed(80 + i))
        cs0 = torch.randn(N1, generator=torch.Generator().manual_seed(95 + i))
        w = (torch.rand(M, generator=torch.Generator().manual_seed(3 + i)) < 0.8).to(torch.bfloat16)
        c, cs = dev(c0), (dev(cs0) if cs_kind else None)
        if cs_kind == "w":
            probs.append((dev(a), dev(b), c, N1, N2, cs, dev(w), 1.25, alpha))
            rcs = cs0.double() + 1.25 * (w.double()[:, None] * a[:, :N1].double()).sum(0)
        else:
            probs.append((dev(a), dev(b), c, N1, N2, cs, None, 1.0, alpha))
            rcs = cs0.double() + a[:, :N1].double().sum(0)
        refs.append((c0.double() + alpha * (a[:, :N1].double().t() @ b[:, :N2].double()), rcs))
    ops.gemm_tn_acc_grouped(probs)
    for (q, (rc, rcs)) in zip(probs, refs):
        assert rel(q[2], rc) < TOL_F32, rel(q[2], rc)
        if q[5] is not None:
            assert rel(q[5], rcs) < TOL_F32, rel(q[5], rcs)


def test_gemm_tn_8p_launch_of_many_blocks_is_unsplit_and_exact(ops):
    """the weight gradients of several blocks in ONE launch (functional's weight-gradient window): about one 192 x 192 tile per CU, so no
    token axis is cut and the tiles are plain read-add-stores -- every result is bitwise reproducible and matches fp64; two problems
    that add into the SAME gradient (a weight used twice) keep their atomics; a problem the tile kernel does not take (486 wide)
    and twelve LayerNorm reductions ride along"""
    M = 4096
    specs = [(384, 576)] * 20 + [(768, 384)] * 4           # 20 * 6 + 4 * 8 = 152 tiles
    ops_a = [dev(rnd(M, 768, seed=300 + i)) for i in range(3)]
    ops_b = [dev(rnd(M, 576, seed=310 + i)) for i in range(3)]
    def build():
        probs = []
        for i, (n1, n2) in enumerate(specs):
            c = torch.full((n1, n2), 0.25 * (i % 3), device="cuda")
            cs = torch.zeros(n1, device="cuda") if i % 2 else None
            probs.append((ops_a[i % 3], ops_b[(i + 1) % 3], c, n1, n2, cs, None, 1.0, 1.0 if i % 4 else 0.5))
        shared = torch.zeros(384, 384, device="cuda")
        probs.append((ops_a[0], ops_b[0], shared, 384, 384, None))
        probs.append((ops_a[1], ops_b[1], shared, 384, 384, None))
        probs.append((ops_a[2], ops_b[2], torch.zeros(486, 192, device="cuda"), 486, 192, torch.zeros(486, device="cuda")))
        return probs
    lns = []
    for k in range(12):
        C = (192, 384)[k % 2]
        x, dy, g = rnd(1000, C, seed=400 + k), rnd(1000, C, seed=420 + k), torch.randn(C, generator=torch.Generator().manual_seed(k))
        mean = x.float().mean(-1); rstd = (x.float().var(-1, unbiased=False) + 1e-5).rsqrt()
        lns.append((dev(dy), dev(x), dev(g.float()), dev(mean), dev(rstd)))
    def partials():
        items, outs = [], []
        for dy, x, g, mean, rstd in lns:
            dg, db = torch.zeros_like(g), torch.zeros_like(g)
            ops.layernorm_bwd(dy, x, g, mean, rstd, None, dg, db, defer=items)
            outs.append((dg, db))
        return items, outs
    p0 = build(); it0, out0 = partials()
    ops.gemm_tn_acc_grouped(p0, ln=it0)
    p1 = build(); it1, out1 = partials()
    ops.gemm_tn_acc_grouped(p1, ln=it1)
    for i, (q0, q1) in enumerate(zip(p0, p1)):
        a, b, c, n1, n2 = q0[:5]
        alpha = q0[8] if len(q0) > 8 else 1.0
        if i < len(specs):
            if os.environ.get("AP_GEMM_TN_8P", "1") != "0":
                assert torch.equal(q0[2], q1[2]), i               # unsplit: no atomics, no run-to-run variation
            ref = 0.25 * (i % 3) + alpha * (a[:, :n1].double().t() @ b[:, :n2].double()).cpu()
            assert rel(c, ref) < TOL_F32, (i, rel(c, ref))
            if q0[5] is not None:
                assert rel(q0[5], a[:, :n1].double().sum(0).cpu()) < TOL_F32
    ref = sum((ops_a[k][:, :384].double().t() @ ops_b[k][:, :384].double()).cpu() for k in range(2))
    assert rel(p0[-2][2], ref) < TOL_F32
    ref = (ops_a[2][:, :486].double().t() @ ops_b[2][:, :192].double()).cpu()
    assert rel(p0[-1][2], ref) < TOL_F32 and rel(p0[-1][5], ops_a[2][:, :486].double().sum(0).cpu()) < TOL_F32
    it2, out2 = partials()
    ops.layernorm_bwd_reduce_batched(it2)
    for (g0, b0), (g2, b2) in zip(out0, out2):
        assert rel(g0, g2.cpu()) < 1e-5 and rel(b0, b2.cpu()) < 1e-5 and float(g2.abs().max()) > 0


def test_mhsa_out_row_scale(ops):
    """0/1 DropPath keep mask folded into the attention output: rows of dropped samples are exact zeros, kept ones unchanged;
    the backward of a kept sample is unaffected and a dropped one (dout = 0) yields zeros.  Both kernel families."""
    for (B, N, heads, hd) in [(4, 196, 3, 32), (3, 300, 2, 48)]:
        C = heads * hd
        qkv, do = rnd(B * N, 3 * C, seed=1), rnd(B * N, C, seed=2)
        keep = torch.tensor([1.0, 0.0, 1.0, 1.0][:B])
        scale = hd ** -0.5
        o_ref, lse_ref = ops.mhsa_fwd(dev(qkv), B, N, heads, scale)
        o, lse = ops.mhsa_fwd(dev(qkv), B, N, heads, scale, out_row_scale=keep.cuda())
        assert torch.equal(o.reshape(B, N, C)[[0, 2]], o_ref.reshape(B, N, C)[[0, 2]]) and torch.equal(lse, lse_ref)
        assert float(o.reshape(B, N, C)[1].abs().max()) == 0.0
        do_m = (do.float().reshape(B, N, C) * keep[:, None, None]).to(torch.bfloat16).reshape(B * N, C)      # what the proj dgrad epilogue delivers
        d1 = ops.mhsa_bwd(dev(qkv), o, dev(do_m), lse, B, N, heads, scale)
        d0 = ops.mhsa_bwd(dev(qkv), o_ref, dev(do_m), lse, B, N, heads, scale)
        assert torch.equal(d1, d0) and float(d1.reshape(B, N, 3 * C)[1].abs().max()) == 0.0


def test_colsum(ops):
    for M, N in [(1000, 192), (77, 486), (5000, 1000), (10, 16)]:
        a = rnd(M, ops.round_up(N, 8), seed=M)
        out = torch.zeros(N, device="cuda")
        ops.colsum_acc(dev(a), out)
        assert rel(out, a[:, :N].double().sum(0)) < TOL_F32


@pytest.mark.parametrize("B,H,W,heads", [(2, 8, 8, 2), (2, 7, 7, 2), (1, 6, 10, 2), (2, 5, 9, 1), (3, 28, 28, 6), (1, 16, 16, 3), (2, 20, 20, 1)])
def test_outlook_core(ops, B, H, W, heads):
    C = heads * 32
    h, w = (H + 1) // 2, (W + 1) // 2
    ldl = ops.round_up(heads * 81, 8)
    v = rnd(B, H, W, C, seed=1)
    logits = rnd(B * h * w, ldl, scale=2.0, seed=2)
    dy = rnd(B, H, W, C, seed=3)
    scale = 32 ** -0.5
    vr = v.double().requires_grad_(True)
    lr = logits[:, :heads * 81].double().reshape(B, h, w, heads * 81).requires_grad_(True)
    yr = R.outlook_core(vr, lr, heads)
    yr.backward(dy.double())
    y = ops.outlook_fwd(dev(v), dev(logits), heads, scale)
    assert rel(y, yr) < TOL_BF16
    dv, dl = ops.outlook_bwd(dev(v), dev(logits), dev(dy), heads, scale)
    assert rel(dv, vr.grad) < TOL_BF16
    assert rel(dl[:, :heads * 81], lr.grad.reshape(B * h * w, heads * 81)) < TOL_BF16
    if ldl > heads * 81:
        assert float(dl[:, heads * 81:].float().abs().sum()) == 0.0


@pytest.mark.parametrize("B,H,W,C", [(2, 8, 8, 32), (2, 7, 7, 64), (1, 5, 9, 32), (2, 28, 28, 192)])
def test_avgpool(ops, B, H, W, C):
    x = rnd(B, H, W, C, seed=1)
    xr = x.double().requires_grad_(True)
    pr = R.avgpool_ceil(xr, 2)
    dp = rnd(*pr.shape, seed=2)
    pr.backward(dp.double())
    p = ops.avgpool2_fwd(dev(x))
    assert rel(p, pr) < TOL_BF16
    base = rnd(B, H, W, C, seed=3)
    dx = dev(base).clone()
    ops.avgpool2_bwd_acc(dev(dp), dx)
    assert rel(dx, base.double() + xr.grad) < TOL_BF16


@pytest.mark.parametrize("B,N,heads,hd", [(2, 196, 12, 32), (3, 64, 2, 32), (2, 100, 4, 32), (1, 144, 12, 32), (2, 197, 3, 32), (2, 25, 2, 32),
                                          (1, 256, 2, 32), (5, 36, 1, 32), (2, 197, 3, 64), (2, 64, 2, 64), (1, 256, 1, 64), (3, 50, 6, 64),
                                          # key/query-blocked kernels (mhsa_flash.hip): head_dim 48 of VOLO-D4/D5, N > 256 (448 px -> 784 tokens)
                                          (2, 784, 16, 48), (1, 196, 16, 48), (2, 50, 3, 48), (1, 784, 2, 32), (1, 400, 3, 64), (2, 257, 1, 32),
                                          (1, 130, 2, 48), (3, 64, 1, 48), (1, 1025, 1, 48),
                                          # 276 / 270 items on 256 persistent workgroups (some take two); a full and a one-key last tile
                                          (23, 196, 12, 32), (2, 208, 2, 32), (1, 193, 1, 32), (45, 144, 6, 32)])
def test_mhsa(ops, B, N, heads, hd):
    C = heads * hd
    qkv = rnd(B * N, 3 * C, seed=1)
    do = rnd(B * N, C, seed=2)
    scale = hd ** -0.5
    qr = qkv.double().reshape(B, N, 3 * C).requires_grad_(True)
    orf = R.mhsa_core(qr, heads)
    orf.backward(do.double().reshape(B, N, C))
    o, lse = ops.mhsa_fwd(dev(qkv), B, N, heads, scale)
    assert rel(o, orf.reshape(B * N, C)) < TOL_BF16
    q, k, _ = qkv.double().reshape(B, N, 3, heads, hd).permute(2, 0, 3, 1, 4)
    lse_ref = torch.logsumexp(q @ k.transpose(-1, -2) * scale, dim=-1)
    assert float((lse.cpu().double() - lse_ref).abs().max()) < 2e-3
    dqkv = ops.mhsa_bwd(dev(qkv), o, dev(do), lse, B, N, heads, scale)
    g = qr.grad.reshape(B * N, 3, C)
    d = dqkv.reshape(B * N, 3, C)
    for i, nm in enumerate("qkv"):
        assert rel(d[:, i], g[:, i]) < 1.5e-2, nm


@pytest.mark.parametrize("B,N,heads,hd", [(2, 197, 12, 32), (3, 65, 2, 32), (4, 17, 1, 32), (2, 785, 16, 48), (3, 197, 3, 64), (2, 50, 2, 48)])
def test_class_attention(ops, B, N, heads, hd):
    C = heads * hd
    q, kv, do = rnd(B, C, seed=1), rnd(B * N, 2 * C, seed=2), rnd(B, C, seed=3)
    scale = hd ** -0.5
    qr = q.double().requires_grad_(True)
    kvr = kv.double().reshape(B, N, 2, heads, hd).requires_grad_(True)
    kk, vv = kvr[:, :, 0].transpose(1, 2), kvr[:, :, 1].transpose(1, 2)
    att = torch.softmax((qr.reshape(B, heads, 1, hd) * scale) @ kk.transpose(-1, -2), dim=-1)
    orf = (att @ vv).transpose(1, 2).reshape(B, C)
    orf.backward(do.double())
    o, probs = ops.class_attn_fwd(dev(q), dev(kv), B, N, heads, scale)
    assert rel(o, orf) < TOL_BF16
    assert rel(probs, att.reshape(B, heads, N)) < 1e-3
    dq, dkv = ops.class_attn_bwd(dev(q), dev(kv), probs, dev(do), B, N, heads, scale)
    assert rel(dq, qr.grad) < TOL_BF16
    assert rel(dkv, kvr.grad.reshape(B * N, 2 * C)) < TOL_BF16


def test_mix_token_swap_exact(ops):
    x = rnd(5, 12, 10, 32, seed=1)
    for box in [(1, 2, 4, 5), (0, 0, 0, 0), (0, 0, 6, 5), (3, 1, 6, 2)]:
        y = ops.mix_token_swap(dev(x), 2 * box[0], 2 * box[2], 2 * box[1], 2 * box[3])
        assert torch.equal(y.cpu(), R.mix_token_swap(x, box, 2))


@pytest.mark.parametrize("B,N,C", [(4, 196, 1000), (3, 16, 16), (2, 9, 20), (5, 1, 1000), (2, 36, 12)])
def test_soft_ce(ops, B, N, C):
    ldx = ops.round_up(C, 8)
    g = torch.Generator().manual_seed(B * 31 + N)
    logits = torch.zeros(B * N, ldx, dtype=torch.bfloat16)
    logits[:, :C] = (torch.randn(B * N, C, generator=g) * 2).to(torch.bfloat16)
    target = torch.rand(B, C, 2 + N, generator=g) * (torch.rand(B, C, 2 + N, generator=g) < 0.05) + 0.1 / C
    gs = 0.5 / (B * N)
    xr = logits[:, :C].double().requires_grad_(True)
    t_aux = target[:, :, 2:].transpose(1, 2).reshape(-1, C).double()
    lse = torch.logsumexp(xr, -1, keepdim=True)
    rows = -(t_aux * (xr - lse)).sum(-1)
    (rows.sum() * gs).backward()
    tdev = dev(target)
    row_loss, dl = ops.soft_ce_fwd_bwd(dev(logits), C, tdev[:, :, 2:], tdev.stride(0), tdev.stride(1), tdev.stride(2), N, gs)
    assert rel(row_loss, rows) < 1e-4
    assert rel(dl[:, :C], xr.grad) < TOL_BF16
    if ldx > C:
        assert float(dl[:, C:].float().abs().sum()) == 0.0
    assert abs(float(R.soft_target_ce(xr.detach(), t_aux)) - float(row_loss.double().mean())) < 1e-4


def test_small_elementwise(ops):
    x = rnd(6 * 49, 64, seed=1)
    sc = torch.tensor([0.0, 1.25, 1.25, 0.0, 1.25, 1.25])
    y = ops.row_scale(dev(x), dev(sc), 49)
    assert rel(y, x.double() * sc.double().repeat_interleave(49)[:, None]) < 1e-2
    a, b = rnd(4, 7, 7, 64, seed=2), rnd(1, 7, 7, 64, seed=3)
    assert rel(ops.add_bcast(dev(a), dev(b)), a.double() + b.double()) < 1e-2
    out = torch.zeros(7 * 7 * 64, device="cuda")
    ops.sum_reps_acc(dev(a), out, 4)
    assert rel(out, a.double().sum(0).reshape(-1)) < 1e-3


@pytest.mark.parametrize("B,r_in,r_out", [(3, 224, 128), (2, 224, 160), (2, 224, 192), (2, 224, 224), (2, 64, 96), (1, 37, 20)])
def test_resize_bilinear(ops, B, r_in, r_out):
    """per-step input resize (main_prog.py:973): against torch's own F.interpolate on the CPU (the reference calls exactly that),
    output bf16 NHWC: <= 1 bf16 ulp of the fp32 result"""
    x = torch.randn(B, 3, r_in, r_in, generator=torch.Generator().manual_seed(r_out))
    ref = torch.nn.functional.interpolate(x, size=(r_out, r_out), mode="bilinear", align_corners=False).permute(0, 2, 3, 1)
    y = ops.resize_bilinear_nhwc(dev(x), r_out)
    assert y.shape == (B, r_out, r_out, 3) and y.dtype == torch.bfloat16
    err = (y.float().cpu() - ref).abs()
    assert float((err / (ref.abs() * 2 ** -8 + 1e-6)).max()) <= 1.01, float(err.max())
    if r_in == r_out:
        assert torch.equal(y.cpu(), ref.to(torch.bfloat16))


def test_droppath_masks(ops):
    """all DropPath sites of a forward pass in one launch == timm's floor(keep + U) / keep per site, plus the per-token masks"""
    g = torch.Generator().manual_seed(0)
    sites, B, tokens = 5, 7, 13
    u = torch.rand(sites, B, generator=g)
    keep = torch.tensor([1.0, 0.9, 0.5, 0.97, 0.3])
    f, m, tm = ops.droppath_masks(dev(u), dev(keep), tokens)
    m_ref = torch.floor(keep[:, None] + u)
    assert torch.equal(m.cpu(), m_ref) and torch.allclose(f.cpu(), m_ref / keep[:, None])
    assert tm.shape == (sites, 96) and torch.equal(tm[:, :B * tokens].float().cpu(), m_ref.repeat_interleave(tokens, dim=1))
    assert float(tm[:, B * tokens:].float().abs().sum()) == 0.0
    f2, m2, none = ops.droppath_masks(dev(u), dev(keep), 0)
    assert none is None and torch.equal(m2.cpu(), m_ref)


def test_errors_are_loud(ops):
    from autoprog_amd._lib import AutoProgHipError
    with pytest.raises(AutoProgHipError):
        ops.layernorm_fwd(torch.zeros(4, 64, dtype=torch.bfloat16), torch.ones(64), torch.zeros(64), 1e-5)   # CPU tensor
    with pytest.raises(AutoProgHipError):
        ops.mhsa_fwd(torch.zeros(2 * 30, 3 * 40, dtype=torch.bfloat16, device="cuda"), 2, 30, 1, 40 ** -0.5)   # head_dim 40: unsupported


def test_fused_adamw_ema_matches_torch(ops):
    from autoprog_amd.dist import GradientBucketReducer
    from autoprog_amd.optim import FlatAdamWEma

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.a = torch.nn.Linear(37, 53)
            self.ln = torch.nn.LayerNorm(53)
            self.b = torch.nn.Linear(53, 11)
            self.pos_embed = torch.nn.Parameter(torch.randn(1, 5, 53))

        def no_weight_decay(self):
            return {"pos_embed"}

        def forward(self, x):
            return self.b(self.ln(self.a(x) + self.pos_embed.mean(1)))

    torch.manual_seed(0)
    net, ref = Net().cuda(), Net().cuda()
    ref.load_state_dict(net.state_dict())
    decays = [0.998, 0.9986, 0.999, 0.9996]
    red = GradientBucketReducer(list(net.parameters()), world_size=1)
    opt = FlatAdamWEma(net, red, lr=1.6e-3, weight_decay=0.05, ema_decays=decays)
    dec, nodec = [], []
    for n, p in ref.named_parameters():
        (nodec if (p.dim() == 1 or n.endswith(".bias") or n == "pos_embed") else dec).append(p)
    ropt = torch.optim.AdamW([{"params": dec, "weight_decay": 0.05}, {"params": nodec, "weight_decay": 0.0}], lr=1.6e-3)
    remas = [[p.detach().clone() for p in ref.parameters()] for _ in decays]
    for step in range(4):
        x = torch.randn(16, 37, device="cuda")
        red.zero_grad()
        net(x).pow(2).mean().backward()
        red.finish()
        opt.step()
        ropt.zero_grad()
        ref(x).pow(2).mean().backward()
        ropt.step()
        for d, em in zip(decays, remas):
            torch._foreach_lerp_(em, [p.detach() for p in ref.parameters()], 1.0 - d)
    for (n, p), q in zip(net.named_parameters(), ref.parameters()):
        assert torch.allclose(p, q, atol=2e-6, rtol=1e-5), n
    sd = opt.ema_state_dict(2)
    for (n, _), e in zip(ref.named_parameters(), remas[2]):
        assert torch.allclose(sd[n], e, atol=2e-6, rtol=1e-5), n


@pytest.mark.parametrize("B,H,W,C", [(4, 16, 16, 64), (2, 9, 7, 16), (3, 12, 12, 128), (2, 5, 5, 8)])
def test_bn_relu_fused(ops, B, H, W, C):
    from autoprog_amd import functional as AF
    x = rnd(B, H, W, C, scale=1.5, seed=1) + 0.3
    g = torch.randn(C, generator=torch.Generator().manual_seed(2)) * 0.3 + 1
    b = torch.randn(C, generator=torch.Generator().manual_seed(3)) * 0.3
    dy = rnd(B, H, W, C, seed=4)
    # reference: torch BatchNorm2d (training) + ReLU in fp64 on the NCHW view
    bn = torch.nn.BatchNorm2d(C).double().train()
    with torch.no_grad():
        bn.weight.copy_(g); bn.bias.copy_(b)
    xr = x.double().permute(0, 3, 1, 2).contiguous().requires_grad_(True)
    yr = torch.relu(bn(xr))
    yr.backward(dy.double().permute(0, 3, 1, 2))
    rm, rv = torch.zeros(C, device="cuda"), torch.ones(C, device="cuda")
    xg = dev(x).requires_grad_(True)
    gg, bg = dev(g).requires_grad_(True), dev(b).requires_grad_(True)
    y = AF.BNReLUFn.apply(xg, gg, bg, rm, rv, True, 0.1, 1e-5)
    y.backward(dev(dy))
    assert rel(y, yr.permute(0, 2, 3, 1)) < TOL_BF16
    assert rel(xg.grad, xr.grad.permute(0, 2, 3, 1)) < 1.5e-2
    assert rel(gg.grad, bn.weight.grad) < 5e-3 and rel(bg.grad, bn.bias.grad) < 5e-3
    assert rel(rm, bn.running_mean) < 1e-4 and rel(rv, bn.running_var) < 1e-4
    # eval mode with the running statistics
    bn.eval()
    ye = AF.BNReLUFn.apply(dev(x), dev(g), dev(b), rm, rv, False, 0.1, 1e-5)
    assert rel(ye, torch.relu(bn(x.double().permute(0, 3, 1, 2))).permute(0, 2, 3, 1)) < TOL_BF16
    # (round 5) ap_bn_relu_bwd_act: the same dx / dgamma / dbeta bits, and the activation of the forward pass next to them
    mean = dev(x).float().mean((0, 1, 2))
    rstd = (dev(x).float().var((0, 1, 2), unbiased=False) + 1e-5).rsqrt()
    gd, bd = dev(g).float(), dev(b).float()
    dg0, db0, dg1, db1 = (torch.zeros(C, device="cuda") for _ in range(4))
    dx0 = ops.bn_relu_bwd(dev(dy), dev(x), gd, bd, mean, rstd, dg0, db0)
    dx1, act = ops.bn_relu_bwd(dev(dy), dev(x), gd, bd, mean, rstd, dg1, db1, act_out=True)
    assert torch.equal(dx0, dx1) and torch.equal(dg0, dg1) and torch.equal(db0, db1)
    rme, rve = mean.clone(), (1.0 / rstd ** 2 - 1e-5)
    ya, _, _ = ops.bn_relu_fwd(dev(x), gd, bd, rme, rve, False, 0.1, 1e-5)                 # eval mode: mean / rstd as given
    assert float((act.float() - ya.float()).abs().max()) <= 2 ** -7 * float(ya.float().abs().max())    # (rstd through 1 / sqrt(var + eps): <= 1 bf16 step)
    sc = rstd * gd
    ref = torch.clamp_min(torch.addcmul(bd - mean * sc, dev(x).float(), sc), 0).to(torch.bfloat16)     # the association of k_bn_relu_apply (fma contraction aside)
    assert (act != ref).float().mean() < 2e-3 and float((act.float() - ref.float()).abs().max()) <= 2 ** -7 * float(ref.float().abs().max())


def test_grouped_wgrad_launch_carries_the_layernorm_reductions(ops):
    """ap_gemm_tn_acc_grouped_ln: the deferred dgamma / dbeta reductions of a block done by extra workgroups of the weight-gradient
    launch equal the stand-alone batched reduction (same partial rows; only the order of the fp32 additions differs) and the weight
    gradients themselves are untouched"""
    torch.manual_seed(5)
    rows = 3000
    lns, refs = [], []
    for C in (192, 384, 96):
        x, dy, g = rnd(rows, C, seed=C), rnd(rows, C, seed=C + 1), torch.randn(C)
        mean = x.float().mean(-1); rstd = (x.float().var(-1, unbiased=False) + 1e-5).rsqrt()
        lns.append((dev(dy), dev(x), dev(g.float()), dev(mean), dev(rstd)))
    def partials():
        items, outs = [], []
        for dy, x, g, mean, rstd in lns:
            dg, db = torch.zeros_like(g), torch.zeros_like(g)
            ops.layernorm_bwd(dy, x, g, mean, rstd, None, dg, db, defer=items)
            outs.append((dg, db))
        return items, outs
    M = 4096
    a, b = dev(rnd(M, 384, seed=11)), dev(rnd(M, 192, seed=12))
    def problems():
        return [(a, b, torch.zeros(384, 192, device="cuda"), 384, 192, torch.zeros(384, device="cuda")),
                (b, a, torch.zeros(192, 384, device="cuda"), 192, 384, None)]
    it0, out0 = partials(); p0 = problems()
    ops.layernorm_bwd_reduce_batched(it0); ops.gemm_tn_acc_grouped(p0)
    it1, out1 = partials(); p1 = problems()
    ops.gemm_tn_acc_grouped(p1, ln=it1)
    for (g0, b0), (g1, b1) in zip(out0, out1):
        assert rel(g1, g0.cpu()) < 1e-5 and rel(b1, b0.cpu()) < 1e-5
        assert float(g0.abs().max()) > 0
    for q0, q1 in zip(p0, p1):
        assert rel(q1[2], q0[2].cpu()) < 1e-5
    assert rel(p1[0][5], p0[0][5].cpu()) < 1e-5


@pytest.mark.parametrize("env", [{"AP_MHSA_FLASH": "1"}, {"AP_MHSA_BWD_DS": "0", "AP_MHSA_FWD_P": "0"}, {"AP_GEMM_LDS_EPI": "1"}, {"AP_GEMM_LDS_EPI": "0"}, {"AP_OUTLOOK_MFMA": "0"},
                                 {"AP_STEM_HIP_CONV": "0"}, {"AP_ASYNC_WGRAD": "1"}, {"AP_GEMM_TN_PLACE": "0"}, {"AP_FUSE_LN_REDUCE": "0"}, {"AP_CONV_WGRAD_P": "0"},
                                 {"AP_GEMM_8P": "0"}, {"AP_GEMM_8P": "2"}, {"AP_GEMM_TN_8P": "0"}, {"AP_GELU_STORE_GRAD": "0"}, {"AP_GELU_STORE_GRAD": "1"}, {"AP_LN_BWD_PF": "0"}, {"AP_GEMM_BM224": "0"}, {"AP_GEMM_WS": "0"}, {"AP_GELU_TABLE": "0"}, {"AP_FUSE_POOL_BWD": "0"}, {"AP_STEM_FUSE_BN_PROJ": "0"}, {"AP_STEM_FUSE_BN_BWD_STATS": "0"}, {"AP_BN_PROJ_ACT_IN_BWD": "0"}, {"AP_WGRAD_WINDOW": "0"}, {"AP_STEM_FUSE_BN": "0"},
                                 {"AP_OUTLOOK_P": "0"}, {"AP_OUTLOOK_P": "2"}, {"AP_LN_FWD_LP": "0"}, {"AP_CONV_WAVES": "4"},
                                 {"AP_FUSED_MLP": "0"}, {"AP_FUSED_MLP": "2"}, {"AP_FUSED_MLP_LN": "0"}, {"AP_FUSED_MLP_MIN_ROWS": "0"}, {"AP_MLP_FUSED_V": "1"}])
def test_experimental_kernel_paths_stay_parity_green(env):
    """the kernels kept behind environment switches (DESIGN.md 'What bounds the GEMMs') must keep computing the same thing:
    re-run the GEMM / block tests in a child process with the switch set (the switches are read once per process)"""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    key = next(iter(env))
    if key == "AP_FUSED_MLP_MIN_ROWS":     # the fused MLP in blocks of every size it takes: the batch-8 stage (9, 128 px) has 512 rows per block
        sel, files = "test_d1_train_step_loss_and_every_gradient_vs_oracle", ["tests/test_gpu_fullsize.py"]
    elif key in ("AP_FUSED_MLP", "AP_FUSED_MLP_LN"):          # the transformer blocks' MLP as two launches per direction / fused in the forward only, inside the batch-128 training step
        sel, files = "slice_loss", ["tests/test_gpu_fullsize.py"]
    elif key == "AP_MLP_FUSED_V":      # the one-wave-per-SIMD version of that kernel: bit-identical to the two launches as well
        sel, files = "mlp_fused", ["tests/test_gpu_kernels.py"]
    elif "MHSA" in key:
        sel, files = "test_mhsa", ["tests/test_gpu_kernels.py"]
    elif "AP_LN_" in key:
        sel, files = "layernorm or ln_", ["tests/test_gpu_kernels.py", "tests/test_gpu_fullsize.py"]
    elif "OUTLOOK" in key or "POOL" in key:
        sel, files = "outlook", ["tests/test_gpu_kernels.py", "tests/test_gpu_blocks.py"]
    elif "CONV_WGRAD" in key or "CONV_WAVES" in key:
        sel, files = "conv3x3 or stem64", ["tests/test_gpu_kernels.py", "tests/test_gpu_blocks.py"]
    elif "STEM" in key:          # AP_STEM_HIP_CONV, AP_STEM_FUSE_BN
        sel, files = "d1_shapes or hip_stem or patch_embed", ["tests/test_gpu_model.py", "tests/test_gpu_blocks.py"]
    elif "WGRAD" in key or "FUSE_LN" in key or ("GELU" in key and "TABLE" not in key):
        sel, files = "vs_reference_golden or grad_sink", ["tests/test_gpu_blocks.py", "tests/test_gpu_model.py"]
    else:
        sel, files = "gemm and not experimental", ["tests/test_gpu_kernels.py"]
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "-k", sel] + files, cwd=root, env=dict(os.environ, **env),
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-1000:]


@pytest.mark.parametrize("h,w,h0,w0,C", [(14, 14, 8, 8, 384), (14, 14, 10, 10, 384), (14, 14, 12, 12, 384), (14, 14, 20, 17, 64), (28, 28, 14, 14, 768), (7, 9, 9, 5, 40)])
def test_pos_embed_bicubic_interpolation_vs_torch(ops, h, w, h0, w0, C):
    """functional.PosEmbedInterpFn / ap_resample_grid (VOLO.interpolate_pos_encoding, models/volo.py:580-596) against autograd through
    torch.nn.functional.interpolate(..., mode="bicubic") in fp64 on the CPU: interpolated grid and the gradient of the embedding"""
    import torch.nn.functional as F
    from autoprog_amd import functional as AF
    g = torch.Generator().manual_seed(h * 100 + h0)
    pos = torch.randn(1, h, w, C, generator=g)
    dy = torch.randn(1, h0, w0, C, generator=g)
    p64 = pos.double().requires_grad_(True)
    ref = F.interpolate(p64.permute(0, 3, 1, 2), scale_factor=((h0 + 0.1) / h, (w0 + 0.1) / w), mode="bicubic").permute(0, 2, 3, 1)
    ref.backward(dy.double())
    pg = pos.cuda().requires_grad_(True)
    out = AF.PosEmbedInterpFn.apply(pg, h0, w0)
    assert out.shape == (1, h0, w0, C) and out.dtype == torch.float32
    out.backward(dy.cuda())
    assert rel(out, ref.detach()) < 1e-5, rel(out, ref.detach())
    assert rel(pg.grad, p64.grad) < 1e-5, rel(pg.grad, p64.grad)
    # accumulate = 1 adds to what is there; raw-ABI argument checks
    before = pg.grad.clone()
    out2 = AF.PosEmbedInterpFn.apply(pg, h0, w0)
    out2.backward(dy.cuda())
    assert rel(pg.grad, 2 * before.cpu()) < 1e-6
    from autoprog_amd._lib import lib
    st = torch.cuda.current_stream().cuda_stream
    assert lib.ap_resample_grid(None, h, w, None, None, None, h0, w0, C, 0, st) == -4
    assert lib.ap_resample_grid(out.data_ptr(), 0, w, out.data_ptr(), out.data_ptr(), pg.data_ptr(), h0, w0, C, 0, st) == -1


def test_c_abi_error_codes_and_empty_inputs(ops):
    """the raw C ABI: negative codes instead of exceptions or crashes for null pointers (-4), shape / stride violations (-1),
    configurations the gfx950 kernels do not cover (-2); empty problems are accepted where the header says so"""
    import ctypes
    from autoprog_amd._lib import lib, TnProblem
    x = torch.zeros(64, 64, dtype=torch.bfloat16, device="cuda")
    f = torch.zeros(64, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    P = lambda t: t.data_ptr()
    eps = ctypes.c_float(1e-5)
    assert lib.ap_error_string(-1) and lib.ap_error_string(-2) and lib.ap_error_string(-4) and lib.ap_error_string(0) == b"ok"
    # LayerNorm
    assert lib.ap_layernorm_fwd(None, P(f), P(f), P(x), P(f), P(f), 64, 64, eps, st) == -4
    assert lib.ap_layernorm_fwd(P(x), P(f), P(f), P(x), P(f), P(f), 64, 60, eps, st) == -1          # C not a multiple of 8
    assert lib.ap_layernorm_fwd(P(x), P(f), P(f), P(x), P(f), P(f), 1, 4096, eps, st) == -2          # C > 2048
    assert lib.ap_layernorm_fwd(P(x), P(f), P(f), P(x), P(f), P(f), 0, 64, eps, st) == 0             # no rows: nothing to do
    # GEMMs
    assert lib.ap_gemm_nt(P(x), 64, None, 64, P(x), 64, 64, 64, 64, None, st) == -4
    assert lib.ap_gemm_nt(P(x), 64, P(x), 64, P(x), 64, 64, 64, 60, None, st) == -1                  # K not a multiple of 8
    assert lib.ap_gemm_nt(P(x), 32, P(x), 64, P(x), 64, 64, 64, 64, None, st) == -1                  # lda < K
    assert lib.ap_gemm_nt(P(x), 64, P(x), 64, P(x), 64, 0, 64, 64, None, st) == -1                   # empty M is a caller error here
    c = torch.zeros(64, 64, device="cuda")
    assert lib.ap_gemm_tn_acc(P(x), 64, P(x), 64, P(c), 32, 64, 64, 64, None, st) == -1              # ldc < N2
    assert lib.ap_gemm_tn_acc_grouped(None, 1, None, 0, st) == -4
    arr = (TnProblem * 1)()
    arr[0].A, arr[0].lda, arr[0].B, arr[0].ldb, arr[0].C, arr[0].ldc = P(x), 64, P(x), 64, P(c), 64
    arr[0].M, arr[0].N1, arr[0].N2, arr[0].alpha, arr[0].colsum_A, arr[0].colsum_weight, arr[0].colsum_scale = 64, 64, 64, 1.0, None, None, 1.0
    assert lib.ap_gemm_tn_acc_grouped(ctypes.cast(arr, ctypes.c_void_p), 0, None, 0, st) == -1      # empty group
    assert lib.ap_gemm_tn_acc_grouped(ctypes.cast(arr, ctypes.c_void_p), 33, None, 0, st) == -1     # > AP_TN_MAX_GROUP
    assert lib.ap_gemm_tn_acc_grouped(ctypes.cast(arr, ctypes.c_void_p), 1, None, 0, st) == 0
    assert lib.ap_gemm_tn_grouped_workspace(ctypes.cast(arr, ctypes.c_void_p), 1) == 64 * 64 * 4          # one split, no column sum
    assert lib.ap_gemm_tn_acc_grouped(ctypes.cast(arr, ctypes.c_void_p), 1, P(c), 16, st) == -1           # deterministic mode: workspace too small
    # attention
    q = torch.zeros(2 * 16, 3 * 48, dtype=torch.bfloat16, device="cuda")
    assert lib.ap_mhsa_fwd(P(q), P(q), P(f), 2, 16, 1, 40, ctypes.c_float(0.1), None, st) == -2         # head_dim 40
    assert lib.ap_mhsa_bwd_workspace(2, 196, 12, 32) == 0 and lib.ap_mhsa_bwd_workspace(2, 784, 16, 48) == 2 * 16 * 784 * 4
    assert lib.ap_mhsa_bwd(P(q), P(q), P(q), P(f), P(q), 1, 300, 1, 32, ctypes.c_float(0.1), None, 0, st) == -4   # blocked path without workspace
    assert lib.ap_mhsa_fwd(P(q), None, P(f), 2, 16, 1, 32, ctypes.c_float(0.1), None, st) == -4
    assert lib.ap_outlook_fwd(P(x), P(x), 88, P(x), 1, 8, 8, 1, 16, ctypes.c_float(0.25), st) == -2     # outlook head_dim != 32
    # the fused MLP (ap_mlp_fused_args)
    from autoprog_amd._lib import MlpFusedArgs
    big = torch.zeros(128, 1152, dtype=torch.bfloat16, device="cuda")
    byt = torch.zeros(128, 1152, dtype=torch.uint8, device="cuda")

    def margs(**kw):
        a = MlpFusedArgs()
        a.x, a.ldx, a.wa, a.ldwa, a.wb, a.ldwb, a.out, a.ldo = P(big), 384, P(big), 384, P(big), 1152, P(big), 384
        a.hidden_out, a.ldh, a.codes, a.rows_per_scale, a.m, a.c, a.hidden, a.backward = P(big), 1152, P(byt), 1, 128, 384, 1152, 0
        for k_, v_ in kw.items():
            setattr(a, k_, v_)
        return ctypes.byref(a)
    assert lib.ap_mlp_fused(None, st) == -4
    assert lib.ap_mlp_fused(margs(x=None), st) == -4                                               # no rows and no LayerNorm input
    assert lib.ap_mlp_fused(margs(ldwb=384), st) == -1                                             # ldwb < hidden
    assert lib.ap_mlp_fused(margs(c=256, hidden=768), st) == -2                                    # built for c = 384
    assert lib.ap_mlp_fused(margs(m=100), st) == -2                                                # whole 128-row blocks
    assert lib.ap_mlp_fused(margs(backward=1, bias1=P(f)), st) == -1                               # the backward launch takes no bias
    assert lib.ap_mlp_fused(margs(x=None, ln_in=P(big), ld_ln=384), st) == -4                      # LayerNorm without its outputs / parameters
    lnk = dict(x=None, ln_in=P(big), ld_ln=384, ln_out=P(big), ld_lno=384, ln_gamma=P(f), ln_beta=P(f), ln_eps=1e-5, ln_mean=P(f), ln_rstd=P(f))
    assert lib.ap_mlp_fused(margs(backward=1, **lnk), st) == -1                                    # the LayerNorm belongs to the forward launch
    assert lib.ap_mlp_fused(margs(**dict(lnk, ld_ln=380)), st) == -1
    torch.cuda.synchronize()                                                                          # nothing above may have faulted


# ------------------------------------------------------------------ stem 3x3 convolution (csrc/conv.hip)
@pytest.mark.parametrize("B,H,W", [(2, 20, 37), (3, 33, 16), (1, 7, 5), (4, 112, 112)])
def test_conv3x3_c64_fwd_dgrad_wgrad_vs_torch_fp32(B, H, W):
    """ap_conv3x3_c64 / _wgrad against torch's fp32 convolution on the same bf16-rounded operands (reference stem:
    models/volo.py:359-366, nn.Conv2d(64, 64, 3, 1, 1, bias=False)).  Ragged sizes exercise the tile edges (32 x 16 forward
    tiles, 16 x 16 weight-gradient tiles).  Tolerances: outputs 5e-3 rel-L2 (bf16 store rounding), the partial BatchNorm sums
    1e-5, the fp32 weight gradient 1e-5 (exact fp32 accumulation of bf16 products, summation order aside)."""
    import torch.nn.functional as F
    from autoprog_amd import ops
    torch.manual_seed(B * 1000 + H)
    x = torch.randn(B, H, W, 64, device="cuda").to(torch.bfloat16)
    dy = torch.randn(B, H, W, 64, device="cuda").to(torch.bfloat16)
    w = torch.randn(64, 64, 3, 3, device="cuda") * 0.05
    w16 = w.to(torch.bfloat16).float()
    wf, wb = ops.conv3x3_pack(w)
    y, st = ops.conv3x3_c64(x, wf, True)
    ref = F.conv2d(x.float().permute(0, 3, 1, 2), w16, None, 1, 1).permute(0, 2, 3, 1)
    assert rel(y, ref) < 5e-3
    assert torch.equal(y, ops.conv3x3_c64(x, wf))                         # the statistics epilogue does not change the output
    sums = st.double().sum(0)
    assert rel(sums[0], y.double().sum((0, 1, 2))) < 1e-5 and rel(sums[1], y.double().pow(2).sum((0, 1, 2))) < 1e-5
    dx = ops.conv3x3_c64(dy, wb)
    refdx = F.conv_transpose2d(dy.float().permute(0, 3, 1, 2), w16, None, 1, 1).permute(0, 2, 3, 1)
    assert rel(dx, refdx) < 5e-3
    dw = torch.full((64, 64, 3, 3), 0.5, device="cuda")                   # accumulates (+=)
    ops.conv3x3_c64_wgrad(x, dy, dw)
    refdw = torch.nn.grad.conv2d_weight(x.float().permute(0, 3, 1, 2), (64, 64, 3, 3), dy.float().permute(0, 3, 1, 2), stride=1, padding=1)
    assert rel(dw - 0.5, refdw) < 1e-5
    dw2 = torch.full((64, 64, 3, 3), 0.5, device="cuda")
    ops.conv3x3_c64_wgrad(x, dy, dw2)
    assert torch.equal(dw, dw2)                                           # ordered slab reduction: bit-reproducible


@pytest.mark.parametrize("B,H,W", [(2, 20, 37), (3, 33, 16), (1, 7, 5), (8, 112, 112)])
def test_conv3x3_c64_input_gradient_with_the_batchnorm_backward_sums(B, H, W):
    """ap_conv3x3_c64_bwd_stats: the input-gradient convolution whose epilogue runs the first pass of the BatchNorm + ReLU backward of the
    layer below (models/volo.py:356-366; round 5).  The map it writes is bit-identical to the plain input-gradient launch; its partial
    rows, summed, are the sums k_bn_relu_bwd_reduce forms from that map and z_below (fp64 check 1e-5: fp32 partial sums in another order);
    and ap_bn_relu_bwd_partials on them gives the dx / dgamma / dbeta of ap_bn_relu_bwd (dx: bf16, equal up to the last place where the
    1e-7 difference of the sums crosses a rounding boundary)."""
    from autoprog_amd import ops
    torch.manual_seed(B * 100 + W)
    dz = torch.randn(B, H, W, 64, device="cuda").to(torch.bfloat16)
    zb = (torch.randn(B, H, W, 64, device="cuda") * 1.5 + 0.3).to(torch.bfloat16)
    w = torch.randn(64, 64, 3, 3, device="cuda") * 0.05
    gamma, beta = torch.rand(64, device="cuda") + 0.5, torch.randn(64, device="cuda") * 0.3
    mean = zb.float().mean((0, 1, 2))
    rstd = (zb.float().var((0, 1, 2), unbiased=False) + 1e-5).rsqrt()
    _, wb = ops.conv3x3_pack(w)
    da, part = ops.conv3x3_c64_bwd_stats(dz, wb, zb, (mean, rstd, gamma, beta))
    assert torch.equal(da, ops.conv3x3_c64(dz, wb))
    xh = (zb.double() - mean.double()) * rstd.double()
    m = ((zb.float() - mean) * rstd * gamma + beta > 0).double()          # the kernels' mask, in their arithmetic
    dzm = da.double() * m
    sums = part.double().sum(0)
    assert rel(sums[0], dzm.sum((0, 1, 2))) < 1e-5 and rel(sums[1], (dzm * xh).sum((0, 1, 2))) < 1e-5
    dg0, db0 = torch.full((64,), 0.25, device="cuda"), torch.full((64,), -0.5, device="cuda")
    dg1, db1 = dg0.clone(), db0.clone()
    dx0 = ops.bn_relu_bwd(da, zb, gamma, beta, mean, rstd, dg0, db0)
    dx1 = ops.bn_relu_bwd_partials(da, zb, gamma, beta, mean, rstd, part, dg1, db1)
    assert rel(dg1 - 0.25, dg0 - 0.25) < 1e-5 and rel(db1 + 0.5, db0 + 0.5) < 1e-5
    assert rel(dx1, dx0) < 1e-4 and (dx1 != dx0).float().mean() < 1e-3


@pytest.mark.parametrize("B,H,W", [(2, 20, 37), (3, 33, 16), (1, 7, 5), (2, 64, 48), (5, 56, 56)])
def test_conv3x3_c128_fwd_dgrad_wgrad_vs_torch_fp32(B, H, W):
    """csrc/conv128.hip (ap_conv3x3_c128: the 3x3 convolutions of the 128-wide VOLO-D4 / D5 stem, models/volo.py:359-366 with
    stem_hidden_dim = 128) and ap_conv3x3_c128_wgrad against torch's fp32 convolution on the same bf16-rounded operands.  Ragged sizes
    exercise the 16 x 16 tile edges and workgroups that walk several tiles (the weight-slab ring runs across tiles; (5, 56, 56) is more
    tiles than CUs).  Tolerances as at 64 channels: outputs 5e-3 rel-L2, the partial BatchNorm sums 1e-5, the fp32 weight gradient 1e-5."""
    import torch.nn.functional as F
    from autoprog_amd import ops
    torch.manual_seed(B * 1000 + H)
    x = torch.randn(B, H, W, 128, device="cuda").to(torch.bfloat16)
    dy = torch.randn(B, H, W, 128, device="cuda").to(torch.bfloat16)
    w = torch.randn(128, 128, 3, 3, device="cuda") * 0.04
    w16 = w.to(torch.bfloat16).float()
    wf, wb = ops.conv3x3_pack(w)
    y, st = ops.conv3x3_c64(x, wf, True)
    ref = F.conv2d(x.float().permute(0, 3, 1, 2), w16, None, 1, 1).permute(0, 2, 3, 1)
    assert rel(y, ref) < 5e-3, rel(y, ref)
    assert torch.equal(y, ops.conv3x3_c64(x, wf))                         # the statistics epilogue does not change the output
    sums = st.double().sum(0)
    assert rel(sums[0], y.double().sum((0, 1, 2))) < 1e-5 and rel(sums[1], y.double().pow(2).sum((0, 1, 2))) < 1e-5
    dx = ops.conv3x3_c64(dy, wb)
    refdx = F.conv_transpose2d(dy.float().permute(0, 3, 1, 2), w16, None, 1, 1).permute(0, 2, 3, 1)
    assert rel(dx, refdx) < 5e-3, rel(dx, refdx)
    dw = torch.full((128, 128, 3, 3), 0.5, device="cuda")                 # accumulates (+=)
    ops.conv3x3_c64_wgrad(x, dy, dw)
    refdw = torch.nn.grad.conv2d_weight(x.float().permute(0, 3, 1, 2), (128, 128, 3, 3), dy.float().permute(0, 3, 1, 2), stride=1, padding=1)
    assert rel(dw - 0.5, refdw) < 1e-5, rel(dw - 0.5, refdw)
    dw2 = torch.full((128, 128, 3, 3), 0.5, device="cuda")
    ops.conv3x3_c64_wgrad(x, dy, dw2)
    assert torch.equal(dw, dw2)                                           # ordered slab reduction: bit-reproducible


def test_conv7_s2d_128_output_channels_vs_torch_fp32():
    """the first convolution of the 128-wide stem (nn.Conv2d(3, 128, 7, 2, 3), models/volo.py:355-357 at stem_hidden_dim = 128): the
    64-channel kernel once per channel half with a pixel stride of 128 (ap_conv7_s2d_ld / ap_conv7_s2d_wgrad_ld)"""
    import torch.nn.functional as F
    from autoprog_amd import ops
    torch.manual_seed(7)
    B, R = 3, 48
    img = torch.randn(B, 3, R, R, device="cuda")
    w = torch.randn(128, 3, 7, 7, device="cuda") * 0.1
    xs = ops.resize_bilinear_s2d16(img, R)
    img16 = xs[..., :12].reshape(B, R // 2, R // 2, 2, 2, 3).permute(0, 5, 1, 3, 2, 4).reshape(B, 3, R, R).float()
    y, st = ops.conv7_s2d(xs, ops.conv7_pack(w), True)
    ref = F.conv2d(img16, w.to(torch.bfloat16).float(), None, 2, 3).permute(0, 2, 3, 1)
    assert y.shape == ref.shape == (B, R // 2, R // 2, 128) and rel(y, ref) < 5e-3
    sums = st.double().sum(0)
    assert st.shape[1:] == (2, 128)
    assert rel(sums[0], y.double().sum((0, 1, 2))) < 1e-5 and rel(sums[1], y.double().pow(2).sum((0, 1, 2))) < 1e-5
    dz = torch.randn_like(y)
    dw = torch.full((128, 3, 7, 7), 0.25, device="cuda")
    ops.conv7_s2d_wgrad(xs, dz, dw)
    refdw = torch.nn.grad.conv2d_weight(img16, (128, 3, 7, 7), dz.float().permute(0, 3, 1, 2), stride=2, padding=3)
    assert rel(dw - 0.25, refdw) < 1e-5


def test_conv3x3_bn_relu_fn_vs_torch():
    """Conv3x3BNReLUFn (conv -> BatchNorm(batch stats) -> ReLU) forward / backward against the same triple in torch fp32"""
    import torch.nn.functional as F
    from autoprog_amd import functional as AF
    torch.manual_seed(5)
    B, H, W = 4, 24, 40
    x = torch.randn(B, H, W, 64, device="cuda").to(torch.bfloat16).requires_grad_(True)
    cw = (torch.randn(64, 64, 3, 3, device="cuda") * 0.05).requires_grad_(True)
    g = (1 + 0.1 * torch.randn(64, device="cuda")).requires_grad_(True)
    b = (0.1 * torch.randn(64, device="cuda")).requires_grad_(True)
    rm, rv = torch.zeros(64, device="cuda"), torch.ones(64, device="cuda")
    y = AF.Conv3x3BNReLUFn.apply(x, cw, g, b, rm, rv, True, 0.1, 1e-5)
    dy = torch.randn_like(y)
    y.backward(dy)
    x2 = x.detach().float().permute(0, 3, 1, 2).requires_grad_(True)
    cw2 = cw.detach().to(torch.bfloat16).float().requires_grad_(True)
    g2, b2 = g.detach().clone().requires_grad_(True), b.detach().clone().requires_grad_(True)
    rm2, rv2 = torch.zeros(64, device="cuda"), torch.ones(64, device="cuda")
    z = F.conv2d(x2, cw2, None, 1, 1)
    z = z.detach().to(torch.bfloat16).float() + (z - z.detach())         # the HIP path stores the convolution output in bf16 (value only)
    y2 = F.relu(F.batch_norm(z, rm2, rv2, g2, b2, True, 0.1, 1e-5))
    y2.backward(dy.float().permute(0, 3, 1, 2))
    assert rel(y, y2.permute(0, 2, 3, 1)) < 6e-3
    assert rel(rm, rm2) < 1e-4 and rel(rv, rv2) < 1e-4
    errs = dict(dx=rel(x.grad, x2.grad.permute(0, 2, 3, 1)), dw=rel(cw.grad, cw2.grad), dg=rel(g.grad, g2.grad), db=rel(b.grad, b2.grad))
    print("Conv3x3BNReLUFn gradient rel-L2 errors:", errs)
    assert errs["dx"] < 6e-3 and errs["dw"] < 6e-3 and errs["dg"] < 1e-3 and errs["db"] < 1e-3, errs      # measured 2.3e-3 / 1.7e-3 / 1.5e-6 / 3e-8


@pytest.mark.parametrize("B,H,W,C,N,k", [(2, 16, 24, 64, 192, 4), (3, 28, 28, 192, 384, 2), (1, 8, 8, 32, 20, 2), (4, 112, 112, 64, 192, 4)])
def test_patch_conv_fn_vs_torch_fp32(B, H, W, C, N, k):
    """PatchConvFn (k x k / stride k convolution as patch-addressed GEMMs: PatchEmbed.proj, Downsample -- models/volo.py:368-372,383-396)
    against torch's fp32 convolution on the same bf16-rounded operands: output, input gradient, weight gradient, bias gradient.
    Tolerances: bf16 outputs / input gradient 5e-3 rel-L2, fp32 weight and bias gradients 1e-4 (fp32 atomics: order varies)."""
    import torch.nn.functional as F
    from autoprog_amd import functional as AF
    torch.manual_seed(C + N)
    x = torch.randn(B, H, W, C, device="cuda").to(torch.bfloat16).requires_grad_(True)
    w = (torch.randn(N, C, k, k, device="cuda") * 0.05).requires_grad_(True)
    b = (0.1 * torch.randn(N, device="cuda")).requires_grad_(True)
    assert AF.patch_conv_ok(x, w, k)
    y = AF.PatchConvFn.apply(x, w, b, k)
    dy = torch.randn_like(y)
    y.backward(dy)
    x2 = x.detach().float().permute(0, 3, 1, 2).requires_grad_(True)
    w2 = w.detach().to(torch.bfloat16).float().requires_grad_(True)
    b2 = b.detach().clone().requires_grad_(True)
    y2 = F.conv2d(x2, w2, b2, stride=k)
    y2.backward(dy.float().permute(0, 3, 1, 2))
    assert y.shape == (B, H // k, W // k, N)
    assert rel(y, y2.permute(0, 2, 3, 1)) < 5e-3
    assert rel(x.grad, x2.grad.permute(0, 2, 3, 1)) < 5e-3
    assert rel(w.grad, w2.grad) < 1e-4 and rel(b.grad, b2.grad) < 1e-4


@pytest.mark.parametrize("B,R,src", [(2, 40, 40), (3, 64, 48), (1, 18, 18), (4, 224, 224)])
def test_conv7_s2d_fwd_wgrad_vs_torch_fp32(B, R, src):
    """The first stem convolution (7x7 / stride 2 / pad 3, 3 -> 64: models/volo.py:355-357) on the space-to-depth image:
    ap_resize_bilinear_s2d16 + ap_conv7_s2d / _wgrad against F.interpolate + torch's fp32 convolution on the same bf16-rounded
    operands.  Outputs 5e-3 rel-L2 (bf16 store), partial BatchNorm sums 1e-5, fp32 weight gradient 1e-5."""
    import torch.nn.functional as F
    from autoprog_amd import ops
    torch.manual_seed(R)
    img = torch.randn(B, 3, src, src, device="cuda")
    w = torch.randn(64, 3, 7, 7, device="cuda") * 0.1
    xs = ops.resize_bilinear_s2d16(img, R)
    ref_img = (F.interpolate(img, size=(R, R), mode="bilinear", align_corners=False) if R != src else img).to(torch.bfloat16).float()
    back = xs[..., :12].reshape(B, R // 2, R // 2, 2, 2, 3).permute(0, 5, 1, 3, 2, 4).reshape(B, 3, R, R)        # undo the space-to-depth
    assert rel(back, ref_img) < 4e-3 and float(xs[..., 12:].abs().max()) == 0.0
    img16 = back.float()
    y, st = ops.conv7_s2d(xs, ops.conv7_pack(w), True)
    ref = F.conv2d(img16, w.to(torch.bfloat16).float(), None, 2, 3).permute(0, 2, 3, 1)
    assert y.shape == ref.shape and rel(y, ref) < 5e-3
    sums = st.double().sum(0)
    assert rel(sums[0], y.double().sum((0, 1, 2))) < 1e-5 and rel(sums[1], y.double().pow(2).sum((0, 1, 2))) < 1e-5
    dz = torch.randn_like(y)
    dw = torch.full((64, 3, 7, 7), 0.25, device="cuda")
    ops.conv7_s2d_wgrad(xs, dz, dw)
    refdw = torch.nn.grad.conv2d_weight(img16, (64, 3, 7, 7), dz.float().permute(0, 3, 1, 2), stride=2, padding=3)
    assert rel(dw - 0.25, refdw) < 1e-5


@pytest.mark.parametrize("M,N,K", [(256, 128, 128), (1000, 384, 1152), (4096, 1152, 384), (300, 200, 192), (4352, 384, 1152), (4200, 768, 768), (4100, 3072, 768)])
def test_gemm_nt_fp8_vs_dequantised_reference(M, N, K):
    """ap_quantize_fp8 + ap_gemm_nt_fp8 (BASELINE configs[4] 'mixed MFMA fp8 GEMM'): (a) the quantiser against torch's e4m3 cast of the
    clamped, scaled input (bit exact) and its amax; (b) the GEMM against an fp32 matmul of the SAME dequantised bytes (the only
    differences left are the fp32 summation order and the bf16 store: 4e-3 rel-L2), with bias + GELU + stored pre-activation and
    with a residual; (c) against the un-quantised fp32 product: 6e-2 (e4m3 keeps 3 mantissa bits).  M >= 4096 with K % 128 == 0: the
    fp8 instantiations of the persistent 8-phase kernel (256 x 192 / 256 x 256 tiles), the flavours of a transformer block's forward."""
    import torch.nn.functional as F
    from autoprog_amd import ops
    torch.manual_seed(M + N)
    a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda") * 0.05).to(torch.bfloat16)
    bias = torch.randn(N, device="cuda") * 0.1
    res = torch.randn(M, ops.round_up(N, 8), device="cuda").to(torch.bfloat16)
    amax = torch.zeros(1, device="cuda")
    sa = (ops.FP8_MAX / a.float().abs().amax()).reshape(1)
    a8 = ops.quantize_fp8(a, sa, amax)
    assert float(amax) == float(a.float().abs().amax())
    ref8 = (a.float() * sa).clamp(-448, 448).to(torch.float8_e4m3fn)
    assert torch.equal(a8.view(torch.float8_e4m3fn).float(), ref8.float())
    w8, dq_w = ops.quantize_fp8_now(w)
    dq_a = (1.0 / sa).contiguous()
    a_dq = a8.view(torch.float8_e4m3fn).float() * dq_a
    w_dq = w8.view(torch.float8_e4m3fn).float() * dq_w
    h = torch.empty(M, ops.round_up(N, 8), dtype=torch.bfloat16, device="cuda")
    y = ops.gemm_nt_fp8(a8, w8, dq_a, dq_w, bias=bias, gelu=True, preact_out=h)[:, :N]
    pre = a_dq @ w_dq.t() + bias
    assert rel(h[:, :N], pre) < 4e-3
    assert rel(y, F.gelu(pre.to(torch.bfloat16).float())) < 4e-3
    y2 = ops.gemm_nt_fp8(a8, w8, dq_a, dq_w, residual=res)[:, :N]
    assert rel(y2, a_dq @ w_dq.t() + res[:, :N].float()) < 4e-3
    assert rel(ops.gemm_nt_fp8(a8, w8, dq_a, dq_w)[:, :N], a.float() @ w.float().t()) < 6e-2
    # bias + DropPath row scale + residual (proj / fc2), and GELU with the stored derivative + row scale (fc1)
    rps = max(1, M // 2)
    rs = torch.tensor([0.0, 1.0 / 0.9, 1.0 / 0.9], device="cuda")
    y3 = ops.gemm_nt_fp8(a8, w8, dq_a, dq_w, bias=bias, row_scale=rs, rows_per_scale=rps, residual=res)[:, :N]
    assert rel(y3, pre * rs.repeat_interleave(rps)[:M, None] + res[:, :N].float()) < 4e-3
    gp = torch.empty_like(h)
    y4 = ops.gemm_nt_fp8(a8, w8, dq_a, dq_w, bias=bias, gelu=True, preact_out=gp, preact_grad=True, row_scale=rs, rows_per_scale=rps)[:, :N]
    hb = pre.to(torch.bfloat16).float().requires_grad_(True)
    F.gelu(hb).sum().backward()
    assert rel(gp[:, :N], hb.grad) < 4e-3
    assert rel(y4, F.gelu(hb.detach()) * rs.repeat_interleave(rps)[:M, None]) < 4e-3
    if ops.gemm_nt_fp8_emits(M, N, K):
        # the GELU output a second time as e4m3 (the operand of the next fp8 GEMM): bit-equal to the quantiser run on the bf16 output
        qs, qa, qa_ref = torch.tensor([93.0], device="cuda"), torch.zeros(1, device="cuda"), torch.zeros(1, device="cuda")
        y5, y5_8 = ops.gemm_nt_fp8(a8, w8, dq_a, dq_w, bias=bias, gelu=True, preact_out=gp, preact_grad=True, row_scale=rs, rows_per_scale=rps, q8=(qs, qa))
        assert torch.equal(y5[:, :N], y4)
        assert torch.equal(y5_8[:, :N], ops.quantize_fp8(y5, qs, qa_ref)[:, :N]) and float(qa) == float(y5[:, :N].float().abs().max())


def test_fp8_producers_match_the_quantiser(ops):
    """the two ways an fp8 operand is made without a pass of its own: (a) ap_layernorm_fwd_fp8 -- the LayerNorm's e4m3 copy and amax are
    bit-equal to ap_quantize_fp8 of its bf16 output; (b) ap_quantize_fp8_multi -- several tensors (the Linear weights of a model) in one
    launch, each with its own scale / amax slot, bit-equal to one ap_quantize_fp8 per tensor; (c) the attention kernel's side output"""
    for rows, C in [(777, 384), (300, 768), (64, 192)]:
        x = dev(rnd(rows, C, scale=2.0, seed=rows))
        g = dev(torch.randn(C, generator=torch.Generator().manual_seed(1)) * 0.3 + 1)
        b = dev(torch.randn(C, generator=torch.Generator().manual_seed(2)) * 0.3)
        scale = torch.tensor([37.5], device="cuda")
        amax = torch.zeros(1, device="cuda")
        y, mean, rstd, y8 = ops.layernorm_fwd(x, g, b, 1e-5, fp8=(scale, amax))
        y0, mean0, rstd0 = ops.layernorm_fwd(x, g, b, 1e-5)
        assert torch.equal(y, y0) and torch.equal(mean, mean0) and torch.equal(rstd, rstd0)
        amax_ref = torch.zeros(1, device="cuda")
        assert torch.equal(y8, ops.quantize_fp8(y0, scale, amax_ref)) and float(amax) == float(amax_ref) == float(y0.float().abs().max())
    # (c) ap_mhsa_fwd_fp8: the blocked attention kernel's e4m3 copy of its output (with the DropPath keep mask folded in)
    B, N, heads, hd = 3, 300, 2, 48
    qkv = dev(rnd(B * N, 3 * heads * hd, seed=7))
    keep = torch.tensor([1.0, 0.0, 1.0], device="cuda")
    scale, amax, amax_ref = torch.tensor([55.0], device="cuda"), torch.zeros(1, device="cuda"), torch.zeros(1, device="cuda")
    o, lse, o8 = ops.mhsa_fwd(qkv, B, N, heads, hd ** -0.5, out_row_scale=keep, fp8=(scale, amax))
    o0, lse0 = ops.mhsa_fwd(qkv, B, N, heads, hd ** -0.5, out_row_scale=keep)
    assert torch.equal(o, o0) and torch.equal(lse, lse0)
    assert torch.equal(o8, ops.quantize_fp8(o0, scale, amax_ref)) and float(amax) == float(amax_ref) > 0
    # (d) (round 5) the input gradient of fc2 -- a bf16 launch that multiplies by the 8-bit gelu' codes -- a second time as e4m3: the operand
    # of fc1's fp8 input-gradient product; with and without the DropPath row scale, 256 x 256 and 256 x 192 tiles
    for M, N, K, with_rs in [(4352, 1152, 384, False), (4100, 3072, 768, True), (4096, 576, 192, True)]:
        torch.manual_seed(N)
        g = torch.randn(M, K, device="cuda").to(torch.bfloat16)
        wt = (torch.randn(N, K, device="cuda") * 0.05).to(torch.bfloat16)
        codes = torch.randint(0, 256, (M, N), device="cuda", dtype=torch.uint8)
        rps = max(1, M // 2)
        kw = dict(row_scale=torch.tensor([0.0, 1.0 / 0.9, 1.0 / 0.9], device="cuda"), rows_per_scale=rps) if with_rs else {}
        assert ops.gemm_nt_emits_q8(M, N, K, codes)
        scale, amax, amax_ref = torch.tensor([71.0], device="cuda"), torch.zeros(1, device="cuda"), torch.zeros(1, device="cuda")
        d0 = ops.gemm_nt(g, wt, mul_by=codes, **kw)
        d1, d8 = ops.gemm_nt(g, wt, mul_by=codes, q8=(scale, amax), **kw)
        assert torch.equal(d0, d1)
        assert torch.equal(d8, ops.quantize_fp8(d0, scale, amax_ref)) and float(amax) == float(amax_ref) > 0
    with pytest.raises(Exception):          # refused, not silently skipped: q8 without the codes / on a launch the 8-phase kernel does not take
        ops.gemm_nt(g, wt, row_scale=kw["row_scale"], rows_per_scale=rps, q8=(scale, amax))
    with pytest.raises(Exception):
        ops.gemm_nt(g[:512], wt, mul_by=codes[:512].contiguous(), q8=(scale, amax))
    sizes = [(1152, 384), (384, 384), (64, 16), (3072, 768)]
    ws = [dev(rnd(n, k, scale=0.05 * (i + 1), seed=10 + i)) for i, (n, k) in enumerate(sizes)]
    scales = torch.tensor([100.0, 300.0, 50.0, 1000.0, 7.0], device="cuda")
    amax = torch.zeros(5, device="cuda")
    outs = [torch.empty(w.shape, dtype=torch.uint8, device="cuda") for w in ws]
    slots = [3, 0, 4, 1]
    table = torch.tensor([[w.data_ptr(), o.data_ptr(), w.numel(), s] for w, o, s in zip(ws, outs, slots)], dtype=torch.int64, device="cuda")
    ops.quantize_fp8_multi(table, len(ws), scales, amax)
    for w, o, s in zip(ws, outs, slots):
        a1 = torch.zeros(1, device="cuda")
        assert torch.equal(o, ops.quantize_fp8(w, scales[s:s + 1].contiguous(), a1)) and float(amax[s]) == float(a1)
    assert float(amax[2]) == 0.0


@pytest.mark.parametrize("B,N,heads,hd", [(4, 196, 3, 32), (2, 100, 2, 32), (2, 196, 2, 64), (2, 300, 2, 32), (2, 300, 2, 48), (2, 49, 2, 32), (2, 256, 2, 32)])
def test_mhsa_backward_with_very_negative_logits(ops, B, N, heads, hd):
    """every logit of every row ~ -110, so lse < -88: a padded key (zero K row, s = 0) of the last key tile then has exp(-lse) = inf in
    fp32, and inf times its zero K row put NaN into dQ -- in all three backward kernels, found by a 300-step run on one batch
    (tools/soak.py) once a head's logits had drifted there.  The kernels clamp the exponent (ATT_PCAP); result against fp32 autograd."""
    C = heads * hd
    g = torch.Generator(device="cuda").manual_seed(N + hd)
    w = torch.nn.functional.normalize(torch.randn(1, 1, heads, hd, device="cuda", generator=g), dim=-1) * hd ** 0.5
    c = (110.0 / hd ** 0.5) ** 0.5
    q = c * w + 0.2 * torch.randn(B, N, heads, hd, device="cuda", generator=g)
    k = -c * w + 0.2 * torch.randn(B, N, heads, hd, device="cuda", generator=g)
    v = torch.randn(B, N, heads, hd, device="cuda", generator=g)
    qkv = torch.stack([q, k, v], dim=2).reshape(B * N, 3 * C).to(torch.bfloat16)
    do = (torch.randn(B * N, C, device="cuda", generator=g) * 0.01).to(torch.bfloat16)
    scale = hd ** -0.5
    o, lse = ops.mhsa_fwd(qkv, B, N, heads, scale)
    assert float(lse.max()) < -88.0
    d = ops.mhsa_bwd(qkv, o, do, lse, B, N, heads, scale)
    qf = qkv.float().reshape(B, N, 3, heads, hd).requires_grad_(True)
    S = torch.einsum("bnhd,bmhd->bhnm", qf[:, :, 0], qf[:, :, 1]) * scale
    oref = torch.einsum("bhnm,bmhd->bnhd", torch.softmax(S, -1), qf[:, :, 2]).reshape(B * N, C)
    oref.backward(do.float())
    assert rel(o, oref) < TOL_BF16
    assert bool(torch.isfinite(d.float()).all())
    assert rel(d, qf.grad.reshape(B * N, 3 * C)) < TOL_BF16


def test_kernels_do_not_depend_on_what_lds_held_before(ops):
    """ap_debug_poison_lds fills every CU's LDS with NaN patterns; kernels with padded tiles (196 tokens in 16-row tiles, ragged GEMM
    edges, odd feature maps) must give the same, finite results as on a clean LDS"""
    torch.manual_seed(0)
    cases = {}
    B, N, heads, hd = 4, 196, 6, 32
    qkv = torch.randn(B * N, 3 * heads * hd, device="cuda").to(torch.bfloat16)
    do = torch.randn(B * N, heads * hd, device="cuda").to(torch.bfloat16)
    o, lse = ops.mhsa_fwd(qkv, B, N, heads, hd ** -0.5)
    cases["mhsa_fwd"] = lambda: ops.mhsa_fwd(qkv, B, N, heads, hd ** -0.5)
    cases["mhsa_bwd"] = lambda: (ops.mhsa_bwd(qkv, o, do, lse, B, N, heads, hd ** -0.5),)
    q2 = torch.randn(2 * 300, 3 * 2 * 48, device="cuda").to(torch.bfloat16); d2 = torch.randn(2 * 300, 96, device="cuda").to(torch.bfloat16)
    o2, l2 = ops.mhsa_fwd(q2, 2, 300, 2, 48 ** -0.5)
    cases["flash_fwd"] = lambda: ops.mhsa_fwd(q2, 2, 300, 2, 48 ** -0.5)
    cases["flash_bwd"] = lambda: (ops.mhsa_bwd(q2, o2, d2, l2, 2, 300, 2, 48 ** -0.5),)
    a = torch.randn(4100, 384, device="cuda").to(torch.bfloat16); w = (torch.randn(1152, 384, device="cuda") * 0.05).to(torch.bfloat16)
    cases["gemm_nt_8p"] = lambda: (ops.gemm_nt(a, w),)
    cases["gemm_nt_small"] = lambda: (ops.gemm_nt(a[:300].contiguous(), w),)
    g = torch.randn(4160, 384, device="cuda").to(torch.bfloat16); x = torch.randn(4160, 576, device="cuda").to(torch.bfloat16)
    def tn(gg, xx):
        c = torch.zeros(384, 576, device="cuda"); ops.gemm_tn_acc(gg, xx, c); return (c,)
    cases["gemm_tn_8p"] = lambda: tn(g, x)
    cases["gemm_tn_128"] = lambda: tn(g[:1000].contiguous(), x[:1000].contiguous())
    v = torch.randn(2, 27, 25, 64, device="cuda").to(torch.bfloat16); lg = torch.randn(2 * 14 * 13, 168, device="cuda").to(torch.bfloat16)
    dyo = torch.randn(2, 27, 25, 64, device="cuda").to(torch.bfloat16)
    cases["outlook_fwd"] = lambda: (ops.outlook_fwd(v, lg, 2, 32 ** -0.5),)
    cases["outlook_bwd"] = lambda: ops.outlook_bwd(v, lg, dyo, 2, 32 ** -0.5)
    xc = torch.randn(2, 37, 21, 64, device="cuda").to(torch.bfloat16)
    wf, wb = ops.conv3x3_pack(torch.randn(64, 64, 3, 3, device="cuda") * 0.05)
    cases["conv3x3"] = lambda: (ops.conv3x3_c64(xc, wf),)
    xl = torch.randn(777, 384, device="cuda").to(torch.bfloat16); gl, bl = torch.randn(384, device="cuda"), torch.randn(384, device="cuda")
    cases["layernorm_fwd"] = lambda: ops.layernorm_fwd(xl, gl, bl, 1e-5)
    for name, fn in cases.items():
        ref = fn()
        for pat in (0x7FC07FC0, 0xFFFFFFFF, 0x7F807F80):
            ops.poison_lds(pat)
            out = fn()
            for a_, b_ in zip(out, ref):
                assert bool(torch.isfinite(a_.float()).all()), (name, hex(pat))
                assert torch.allclose(a_.float(), b_.float(), rtol=1e-2, atol=1e-2), (name, hex(pat))
