#!/usr/bin/env python3
"""fp8 (e4m3) forward GEMM vs the bf16 GEMM on the D1 Linear shapes, operands rotated beyond the Infinity Cache.  python tools/bench_fp8.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from autoprog_amd import ops
def timeit(fns, n=40):
    for f in fns[:3]: f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for i in range(n): fns[i % len(fns)]()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
R = 6                                                   # operand copies: > 600 MB in rotation
for (M, N, K, what) in [(12544, 2304, 768, "d5 qkv"), (12544, 3072, 768, "d5 fc1+gelu"), (12544, 768, 3072, "d5 fc2+res"), (12544, 768, 768, "d5 proj+res"), (25088, 1152, 384, "qkv"), (25088, 1152, 384, "fc1+gelu"), (25088, 384, 1152, "fc2+res"), (25088, 384, 384, "proj+res"), (100352, 576, 192, "o.fc1+gelu")]:
    a = [torch.randn(M, K, device="cuda").to(torch.bfloat16) for _ in range(R)]
    w = (torch.randn(N, K, device="cuda") * 0.05).to(torch.bfloat16)
    res = torch.randn(M, N, device="cuda").to(torch.bfloat16)
    h = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    q = [ops.quantize_fp8_now(x) for x in a]
    w8, dqw = ops.quantize_fp8_now(w)
    kw = dict(gelu=True, preact_out=h) if "gelu" in what else (dict(residual=res) if "res" in what else {})
    t16 = timeit([lambda x=x: ops.gemm_nt(x, w, **kw) for x in a])
    t8 = timeit([lambda x8=x8, d=d: ops.gemm_nt_fp8(x8, w8, d, dqw, **kw) for (x8, d) in q])
    one = torch.ones(1, device="cuda")
    tq = timeit([lambda x=x: ops.quantize_fp8(x, one) for x in a])
    fl = 2.0 * M * N * K
    print("%-11s %6dx%4dx%4d  bf16 %6.1f us (%4.0f TF/s)   fp8 %6.1f us (%4.0f TF/s)   x%.2f   | quantise A alone %5.1f us" % (what, M, N, K, t16, fl / t16 / 1e6, t8, fl / t8 / 1e6, t16 / t8, tq))
