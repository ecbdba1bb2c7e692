#!/usr/bin/env python3
"""ap_mlp_fused against the two ap_gemm_nt launches it replaces (bit for bit), and their times over rotating operand sets.
   python tools/check_mlp.py [rows] [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from autoprog_amd import ops  # noqa: E402


def unfused_fwd(x, w1, b1, w2, b2, k2, rs2, n, res):
    h = torch.empty((x.shape[0], w1.shape[0]), dtype=torch.uint8, device=x.device)
    a = ops.gemm_nt(x, w1, bias=b1, gelu=True, preact_out=h, preact_grad=2, row_scale=k2, rows_per_scale=n)
    y = ops.gemm_nt(a, w2, bias=b2, row_scale=rs2, rows_per_scale=n, residual=res)
    return y, a, h


def unfused_bwd(dy, w2t, w1t, codes, rs2, n):
    dh = ops.gemm_nt(dy, w2t, mul_by=codes, row_scale=rs2, rows_per_scale=n)
    dx = ops.gemm_nt(dh, w1t)
    return dx, dh


def main():
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 25088
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    C, H, N = 384, 1152, 196
    dev = "cuda"
    g = torch.Generator().manual_seed(0)
    R = 4 if M >= 16384 else 8

    def rnd(*s, scale=1.0):
        return (torch.randn(*s, generator=g) * scale).to(dev)
    w1 = rnd(H, C, scale=0.05).bfloat16(); w2 = rnd(C, H, scale=0.03).bfloat16()
    b1 = rnd(H, scale=0.1); b2 = rnd(C, scale=0.1)
    w2t = w2.t().contiguous(); w1t = w1.t().contiguous()
    xs = [rnd(M, C).bfloat16() for _ in range(R)]
    ress = [rnd(M, C).bfloat16() for _ in range(R)]
    nb = (M + N - 1) // N
    keep = (torch.rand(nb, generator=g) < 0.9).float().to(dev)
    rs2 = keep / 0.9
    for drop in (False, True):
        k2_, rs2_ = (keep, rs2) if drop else (None, None)
        y0, a0, h0 = unfused_fwd(xs[0], w1, b1, w2, b2, k2_, rs2_, N, ress[0])
        r = ops.mlp_fused(xs[0], w1, w2, bias1=b1, bias2=b2, row_scale_hidden=k2_, row_scale_out=rs2_, rows_per_scale=N, residual=ress[0])
        assert r is not None, "ap_mlp_fused refused the launch"
        y1, a1, h1 = r
        torch.cuda.synchronize()
        da = (a0.float() - a1.float()).abs().max().item(); dh = (h0.int() - h1.int()).abs().max().item()
        dy = (y0.float() - y1.float()).abs().max().item()
        print("forward  drop=%d: a equal %s (max diff %.3g), codes equal %s (%d), out equal %s (max diff %.3g, rel %.2e)" % (
            drop, torch.equal(a0, a1), da, torch.equal(h0, h1), dh, torch.equal(y0, y1), dy,
            ((y0.float() - y1.float()).norm() / y0.float().norm()).item()))
        dyg = xs[1]
        dx0, dh0 = unfused_bwd(dyg, w2t, w1t, h0, rs2_, N)
        r = ops.mlp_fused(dyg, w2t, w1t, backward=True, codes=h0, row_scale_hidden=rs2_, rows_per_scale=N)
        dx1, dh1, _ = r
        torch.cuda.synchronize()
        print("backward drop=%d: dh equal %s (max diff %.3g), dx equal %s (max diff %.3g, rel %.2e)" % (
            drop, torch.equal(dh0, dh1), (dh0.float() - dh1.float()).abs().max().item(), torch.equal(dx0, dx1),
            (dx0.float() - dx1.float()).abs().max().item(), ((dx0.float() - dx1.float()).norm() / dx0.float().norm()).item()))
    # the LayerNorm in front of fc1 inside the launch: against layernorm_fwd + the fused launch, bit for bit
    lg = rnd(C, scale=0.3) + 1.0; lb = rnd(C, scale=0.2)
    xin = (rnd(M, C) * (1.0 + rnd(M, 1).abs()) + rnd(M, 1)).bfloat16()
    xn0, m0_, r0_ = ops.layernorm_fwd(xin, lg, lb, 1e-5)
    yl0, al0, hl0 = ops.mlp_fused(xn0, w1, w2, bias1=b1, bias2=b2, row_scale_hidden=keep, row_scale_out=rs2, rows_per_scale=N, residual=xin)
    r = ops.mlp_fused(None, w1, w2, bias1=b1, bias2=b2, row_scale_hidden=keep, row_scale_out=rs2, rows_per_scale=N, residual=xin, ln=(xin, lg, lb, 1e-5))
    assert r is not None, "ap_mlp_fused refused the LayerNorm launch"
    yl1, al1, hl1, xn1, m1_, r1_ = r
    torch.cuda.synchronize()
    print("forward with LayerNorm: rows equal %s (max diff %.3g), mean equal %s (%.3g), rstd equal %s (%.3g), a equal %s, codes equal %s, out equal %s" % (
        torch.equal(xn0, xn1), (xn0.float() - xn1.float()).abs().max().item(), torch.equal(m0_, m1_), (m0_ - m1_).abs().max().item(),
        torch.equal(r0_, r1_), ((r0_ - r1_).abs() / r0_).max().item(), torch.equal(al0, al1), torch.equal(hl0, hl1), torch.equal(yl0, yl1)))
    # fp64 reference on a few rows (independent of the unfused kernels)
    rows = torch.randint(0, M, (64,), generator=g).to(dev)
    xr = xs[0][rows].double()
    hpre = (xr @ w1.double().t() + b1.double()).bfloat16().double()
    aref = (hpre * 0.5 * (1 + torch.erf(hpre / 2 ** 0.5))) * (keep[rows // N].double()[:, None])
    yref = (aref.bfloat16().double() @ w2.double().t() + b2.double()) * rs2[rows // N].double()[:, None] + ress[0][rows].double()
    print("forward vs fp64 on 64 rows: a %.2e, out %.2e" % (((a1[rows].double() - aref).norm() / aref.norm()).item(), ((y1[rows].double() - yref).norm() / yref.norm()).item()))

    def timeit(fn):
        for i in range(3):
            fn(i)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
        ev[0].record()
        for i in range(reps):
            fn(i)
            ev[i + 1].record()
        torch.cuda.synchronize()
        ts = sorted(ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(reps))
        return ts[len(ts) // 2], ts[0]
    codes = h0
    print("rows %d: fused forward   %.1f us (best %.1f) | unfused %.1f us (best %.1f)" % (
        (M,) + timeit(lambda i: ops.mlp_fused(xs[i % R], w1, w2, bias1=b1, bias2=b2, row_scale_hidden=keep, row_scale_out=rs2, rows_per_scale=N, residual=ress[i % R]))
        + timeit(lambda i: unfused_fwd(xs[i % R], w1, b1, w2, b2, keep, rs2, N, ress[i % R]))))
    def ln_then_fused(i):
        xn, _, _ = ops.layernorm_fwd(xs[i % R], lg, lb, 1e-5)
        return ops.mlp_fused(xn, w1, w2, bias1=b1, bias2=b2, row_scale_hidden=keep, row_scale_out=rs2, rows_per_scale=N, residual=xs[i % R])
    print("rows %d: fused forward with LayerNorm %.1f us (best %.1f) | layernorm_fwd + fused %.1f us (best %.1f)" % (
        (M,) + timeit(lambda i: ops.mlp_fused(None, w1, w2, bias1=b1, bias2=b2, row_scale_hidden=keep, row_scale_out=rs2, rows_per_scale=N, residual=xs[i % R],
                                              ln=(xs[i % R], lg, lb, 1e-5)))
        + timeit(ln_then_fused)))
    print("rows %d: fused backward  %.1f us (best %.1f) | unfused %.1f us (best %.1f)" % (
        (M,) + timeit(lambda i: ops.mlp_fused(xs[i % R], w2t, w1t, backward=True, codes=codes, row_scale_hidden=rs2, rows_per_scale=N))
        + timeit(lambda i: unfused_bwd(xs[i % R], w2t, w1t, codes, rs2, N))))


if __name__ == "__main__":
    main()
