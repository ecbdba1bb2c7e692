#!/usr/bin/env python3
"""NT GEMM kernel choice at the row counts of the early AutoProg stages (8192 = 128 images x 64 tokens at 128 px, 12800 at 160 px, 18432 at
192 px): the persistent 8-phase kernel (AP_GEMM_8P=1, one 256 / 224-row tile per workgroup) against the 128 x 128 / 128 x 64 tile kernels
(AP_GEMM_8P=0) -- run once per setting (the library reads the switch once):  AP_GEMM_8P=0 python tools/sweep_small_m.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from autoprog_amd import ops

def timeit(fns, iters=60):
    for f in fns: f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters): fns[i % len(fns)]()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3

print("AP_GEMM_8P=%s AP_GEMM_BM224=%s" % (os.environ.get("AP_GEMM_8P", "1"), os.environ.get("AP_GEMM_BM224", "1")))
for M in (8192, 12800, 18432):
    for N, K, epi in [(384, 384, "none"), (384, 384, "res"), (384, 1152, "none"), (384, 1152, "res"), (1152, 384, "none"), (1152, 384, "gelu"), (1152, 384, "mul8")]:
        w = (torch.randn(N, K, device="cuda") * K ** -0.5).bfloat16()
        fns = []
        for _ in range(12):
            a = torch.randn(M, K, device="cuda").bfloat16()
            out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
            kw = {}
            if epi == "res":
                kw = dict(bias=torch.randn(N, device="cuda"), residual=torch.randn(M, N, device="cuda").bfloat16(), row_scale=torch.rand(128, device="cuda"), rows_per_scale=M // 128)
            elif epi == "gelu":
                kw = dict(bias=torch.randn(N, device="cuda"), gelu=True, preact_out=torch.empty(M, N, device="cuda", dtype=torch.uint8), preact_grad=2)
            elif epi == "mul8":
                kw = dict(mul_by=torch.randint(0, 256, (M, N), device="cuda", dtype=torch.uint8))
            fns.append(lambda a=a, out=out, kw=kw: ops.gemm_nt(a, w, n=N, k=K, out=out, **kw))
        print("M %6d  N %5d K %5d %-5s  %7.1f us" % (M, N, K, epi, timeit(fns)))
