#!/usr/bin/env python3
"""event-timed outlook kernels.  Three regimes (the 12.7-us gap between the stand-alone and the in-step time of the backward):
   warm  -- ONE operand set, launches separated by Python (what rounds 4-5 quoted as "stand-alone": operands served from the 256 MiB Infinity Cache, the
            chip at its boost clock between launches)
   cold  -- operand sets rotated over > 600 MB (every launch reads from memory, as in the step)
   load  -- cold, and in front of every timed launch ~2 ms of the step's own GEMM (the chip then holds the clock it holds in the step)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from autoprog_amd import ops

B, H, W, heads = 128, 28, 28, 6
C = heads * 32
R = 6
g = torch.Generator().manual_seed(0)
vs = [torch.randn(B, H, W, C, generator=g).cuda().bfloat16() for _ in range(R)]
dys = [torch.randn(B, H, W, C, generator=g).cuda().bfloat16() for _ in range(R)]
lgs = [torch.randn(B * 14 * 14, ops.round_up(heads * 81, 8), generator=g).cuda().bfloat16() for _ in range(R)]
ga = torch.randn(25088, 384, generator=g).cuda().bfloat16()
gw = (torch.randn(1152, 384, generator=g) * 0.05).cuda().bfloat16()


def t(fn, n=24, rotate=False, load=False):
    for i in range(3):
        fn(i % R if rotate else 0)
    ts = []
    for i in range(n):
        if load:
            for _ in range(60):
                ops.gemm_nt(ga, gw)
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        fn(i % R if rotate else 0)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


fwd = lambda i: ops.outlook_fwd(vs[i], lgs[i], heads, 32 ** -0.5)
bwd = lambda i: ops.outlook_bwd(vs[i], lgs[i], dys[i], heads, 32 ** -0.5)
for name, kw in (("warm", {}), ("cold", dict(rotate=True)), ("load", dict(rotate=True, load=True))):
    print("%-5s fwd %.1f us   bwd(dV+dlogits) %.1f us" % (name, t(fwd, **kw), t(bwd, **kw)), flush=True)
