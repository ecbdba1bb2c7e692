#!/usr/bin/env python3
"""event-timed outlook kernels (pre-allocated outputs are not possible through ops, so time a batch of calls)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from autoprog_amd import ops
B, H, W, heads = 128, 28, 28, 6
C = heads * 32
v = torch.randn(B, H, W, C, device="cuda").bfloat16()
dy = torch.randn(B, H, W, C, device="cuda").bfloat16()
logits = torch.randn(B * 14 * 14, ops.round_up(heads * 81, 8), device="cuda").bfloat16()
def t(fn, n=20):
    for _ in range(3): fn()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
print("SR=%s SRW=%s  fwd %.1f us   bwd(dV+dlogits) %.1f us" % (os.environ.get("AP_OUTLOOK_SR", "-"), os.environ.get("AP_OUTLOOK_SRW", "-"),
      t(lambda: ops.outlook_fwd(v, logits, heads, 32 ** -0.5)), t(lambda: ops.outlook_bwd(v, logits, dy, heads, 32 ** -0.5))))
