#!/usr/bin/env python3
"""the key/query-blocked attention kernels (csrc/mhsa_flash.hip) on the VOLO-D5 448-px shape and two others; AP_FLASH_RT=1 | 2 picks the
number of 16-token tiles a wave owns.  usage: [AP_FLASH_RT=1] python tools/bench_flash.py [B]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from autoprog_amd import ops

def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

B0 = int(sys.argv[1]) if len(sys.argv) > 1 else 16
print("AP_FLASH_RT =", os.environ.get("AP_FLASH_RT", "(default 2)"))
for (B, N, H, hd) in [(B0, 784, 16, 48), (B0, 784, 12, 32), (4 * B0, 196, 16, 48)]:
    os.environ["AP_MHSA_FLASH"] = "1"
    torch.manual_seed(0)
    C = H * hd
    qkv = (torch.randn(B * N, 3 * C, device="cuda") * 0.5).to(torch.bfloat16)
    o, lse = ops.mhsa_fwd(qkv, B, N, H, hd ** -0.5)
    do = torch.randn_like(o)
    tf = timeit(lambda: ops.mhsa_fwd(qkv, B, N, H, hd ** -0.5))
    tb = timeit(lambda: ops.mhsa_bwd(qkv, o, do, lse, B, N, H, hd ** -0.5))
    fl = 4.0 * B * H * N * N * hd
    print("B %3d N %4d heads %2d hd %2d: forward %7.1f us (%5.0f TFLOP/s)   backward (delta + dK,dV + dQ) %7.1f us (%5.0f TFLOP/s)" %
          (B, N, H, hd, tf, fl / tf * 1e-6, tb, 2.5 * fl / tb * 1e-6))
