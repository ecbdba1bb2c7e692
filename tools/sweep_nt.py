#!/usr/bin/env python3
"""time ap_gemm_nt on a few shapes for the tile variant forced by AP_GEMM_NT_TILE (one process per variant)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from autoprog_amd import ops
shapes = [(8192, 8192, 8192), (4096, 4096, 4096), (25088, 1152, 384), (25088, 384, 1152), (25088, 384, 384), (100352, 576, 192), (100352, 192, 576)]
if len(sys.argv) > 1:
    shapes = [tuple(int(v) for v in s.split("x")) for s in sys.argv[1:]]
res = []
for M, N, K in shapes:
    a = torch.randn(M, K, device="cuda").bfloat16(); w = (torch.randn(N, K, device="cuda") * K ** -0.5).bfloat16()
    out = torch.empty(M, ops.round_up(N, 8), device="cuda", dtype=torch.bfloat16)
    for _ in range(3):
        ops.gemm_nt(a, w, n=N, k=K, out=out)
    it = 10 if M * N * K > 1e11 else 30
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        ops.gemm_nt(a, w, n=N, k=K, out=out)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / it
    ref = (a.float() @ w.float().t())
    err = float((out[:, :N].float() - ref).norm() / ref.norm())
    res.append("%dx%dx%d %.1fus %.0fTF err %.1e" % (M, N, K, us, 2.0 * M * N * K / us / 1e6, err))
print("tile=%s | " % os.environ.get("AP_GEMM_NT_TILE", "auto") + " | ".join(res))
