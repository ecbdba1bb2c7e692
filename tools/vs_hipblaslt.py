#!/usr/bin/env python3
"""calibration only: the vendor library (torch.matmul -> hipBLASLt / rocBLAS) on the plain D1 GEMM shapes, rotating operand
sets beyond the Infinity Cache, next to ap_gemm_nt.  Not used by the product."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from autoprog_amd import ops
shapes = [(25088, 1152, 384), (25088, 384, 1152), (25088, 384, 384), (100352, 192, 192), (100352, 576, 192), (100352, 192, 576), (8192, 8192, 8192)]
def t(fns, iters=30):
    for f in fns[:3]: f()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for i in range(iters): fns[i % len(fns)]()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters
for M, N, K in shapes:
    nset = max(1, min(16, int(600e6 // (2.0 * (M * K + M * N))) + 1))
    A = [torch.randn(M, K, device="cuda").bfloat16() for _ in range(nset)]
    O = [torch.empty(M, N, device="cuda", dtype=torch.bfloat16) for _ in range(nset)]
    w = (torch.randn(N, K, device="cuda") * K ** -0.5).bfloat16()
    wt = w.t().contiguous()
    lib = t([(lambda a=a, o=o: torch.matmul(a, w.t(), out=o)) for a, o in zip(A, O)])
    lib2 = t([(lambda a=a, o=o: torch.matmul(a, wt, out=o)) for a, o in zip(A, O)])
    mine = t([(lambda a=a, o=o: ops.gemm_nt(a, w, n=N, k=K, out=o)) for a, o in zip(A, O)])
    fl = 2.0 * M * N * K
    print("%7d %5d %5d   library (B as [N,K]^T) %7.1f us %6.0f TF | (B as [K,N]) %7.1f us %6.0f TF | ap_gemm_nt %7.1f us %6.0f TF" % (M, N, K, lib, fl / lib / 1e6, lib2, fl / lib2 / 1e6, mine, fl / mine / 1e6))
