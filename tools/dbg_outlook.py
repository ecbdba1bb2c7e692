#!/usr/bin/env python3
"""outlook kernels against the fp64 oracle, per output and shape (AP_OUTLOOK_P selects the kernel family)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from autoprog_amd import ops
from oracle import ref_cpu as R


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


for (B, H, W, heads) in [(2, 8, 8, 2), (2, 7, 7, 2), (1, 6, 10, 2), (2, 5, 9, 1), (3, 28, 28, 6), (1, 16, 16, 3), (2, 20, 20, 1), (1, 56, 56, 2)]:
    C = heads * 32
    h, w = (H + 1) // 2, (W + 1) // 2
    ldl = ops.round_up(heads * 81, 8)
    g = torch.Generator().manual_seed(1)
    v = torch.randn(B, H, W, C, generator=g).bfloat16()
    logits = (torch.randn(B * h * w, ldl, generator=g) * 2).bfloat16()
    dy = torch.randn(B, H, W, C, generator=g).bfloat16()
    vr = v.double().requires_grad_(True)
    lr = logits[:, :heads * 81].double().reshape(B, h, w, heads * 81).requires_grad_(True)
    yr = R.outlook_core(vr, lr, heads)
    yr.backward(dy.double())
    y = ops.outlook_fwd(v.cuda(), logits.cuda(), heads, 32 ** -0.5)
    dv, dl = ops.outlook_bwd(v.cuda(), logits.cuda(), dy.cuda(), heads, 32 ** -0.5)
    lg = lr.grad.reshape(B * h * w, heads * 81)
    e = (dl[:, :heads * 81].double().cpu() - lg).reshape(B, h, w, heads, 81)
    worst = e.abs().amax(dim=(0, 3, 4))
    print("P=%s %s  y %.2e  dv %.2e  dl %.2e  pad %.1f" % (os.environ.get("AP_OUTLOOK_P", "-"), (B, H, W, heads), rel(y, yr), rel(dv, vr.grad),
          rel(dl[:, :heads * 81], lg), float(dl[:, heads * 81:].float().abs().sum())))
    if rel(dl[:, :heads * 81], lg) > 1e-2:
        print("   worst |err| per window position:\n", (worst * 100).round().int())
