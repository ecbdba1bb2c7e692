#!/usr/bin/env python3
"""run one gemm_nt shape repeatedly (for rocprofv3 --pmc): python tools/one_gemm.py M N K [iters]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from autoprog_amd import ops
M, N, K = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 20
kind = sys.argv[5] if len(sys.argv) > 5 else "nt"
if kind == "nt":
    a = torch.randn(M, K, device="cuda").bfloat16(); w = (torch.randn(N, K, device="cuda") * K ** -0.5).bfloat16()
    out = torch.empty(M, ops.round_up(N, 8), device="cuda", dtype=torch.bfloat16)
    for _ in range(iters):
        ops.gemm_nt(a, w, n=N, k=K, out=out)
else:
    a = torch.randn(M, N, device="cuda").bfloat16(); b = torch.randn(M, K, device="cuda").bfloat16()
    c = torch.zeros(N, K, device="cuda")
    for _ in range(iters):
        ops.gemm_tn_acc(a, b, c)
torch.cuda.synchronize()
