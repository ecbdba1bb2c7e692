#!/usr/bin/env python3
"""ap_mlp_fused against the two launches with COLD weights: 16 weight sets in rotation and a 1-GB copy between launches (the step reads each block's weights
once per direction, from memory; tools/check_mlp.py re-reads one set from L2).  python tools/check_mlp_cold.py [rows]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from autoprog_amd import ops  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 25088
C, H, N, R = 384, 1152, 196, 16
g = torch.Generator().manual_seed(0)
rnd = lambda *s, scale=1.0: (torch.randn(*s, generator=g) * scale).cuda()
w1 = [rnd(H, C, scale=0.05).bfloat16() for _ in range(R)]
w2 = [rnd(C, H, scale=0.03).bfloat16() for _ in range(R)]
w1t = [w.t().contiguous() for w in w1]
w2t = [w.t().contiguous() for w in w2]
b1, b2 = rnd(H, scale=0.1), rnd(C, scale=0.1)
xs = [rnd(M, C).bfloat16() for _ in range(4)]
codes = torch.randint(0, 255, (M, H), dtype=torch.uint8, device="cuda")
flush_a = torch.empty(256 << 20, dtype=torch.float32, device="cuda")
flush_b = torch.empty_like(flush_a)


def timed(fn, flush):
    ts = []
    for i in range(20):
        if flush:
            flush_b.copy_(flush_a)
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); fn(i); e1.record()
        torch.cuda.synchronize()
        if i >= 4:
            ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


def unf_f(i):
    h = torch.empty((M, H), dtype=torch.uint8, device="cuda")
    a = ops.gemm_nt(xs[i % 4], w1[i % R], bias=b1, gelu=True, preact_out=h, preact_grad=2)
    return ops.gemm_nt(a, w2[i % R], bias=b2, residual=xs[(i + 1) % 4])


def unf_b(i):
    dh = ops.gemm_nt(xs[i % 4], w2t[i % R], mul_by=codes)
    return ops.gemm_nt(dh, w1t[i % R])


fus_f = lambda i: ops.mlp_fused(xs[i % 4], w1[i % R], w2[i % R], bias1=b1, bias2=b2, residual=xs[(i + 1) % 4])
fus_b = lambda i: ops.mlp_fused(xs[i % 4], w2t[i % R], w1t[i % R], backward=True, codes=codes)
for flush in (False, True):
    print("%s: forward fused %.1f us | two launches %.1f us;  backward fused %.1f us | two launches %.1f us" % (
        "weights rotated over 16 sets, 1-GB copy in front of every launch" if flush else "weights rotated over 16 sets",
        timed(fus_f, flush), timed(unf_f, flush), timed(fus_b, flush), timed(unf_b, flush)), flush=True)
