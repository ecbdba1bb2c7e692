#!/usr/bin/env python3
"""stem conv 7x7/s2 (3 -> 64): HIP space-to-depth kernels vs torch/MIOpen -- time.  python tools/bench_conv7.py [B] [R]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from autoprog_amd import ops
torch.backends.cudnn.benchmark = True
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
R = int(sys.argv[2]) if len(sys.argv) > 2 else 224
img = torch.randn(B, 3, R, R, device="cuda")
w = torch.randn(64, 3, 7, 7, device="cuda") * 0.1
wp = ops.conv7_pack(w)
xs = ops.resize_bilinear_s2d16(img, R)
print("resize+s2d %.1f us | plain resize %.1f us" % (timeit(lambda: ops.resize_bilinear_s2d16(img, R)), timeit(lambda: ops.resize_bilinear_nhwc(img, R))))
print("HIP conv7 fwd %.1f us | fwd+stats %.1f us" % (timeit(lambda: ops.conv7_s2d(xs, wp)), timeit(lambda: ops.conv7_s2d(xs, wp, True))))
dz = torch.randn(B, R // 2, R // 2, 64, device="cuda").to(torch.bfloat16)
dw = torch.zeros(64, 3, 7, 7, device="cuda")
print("HIP conv7 wgrad %.1f us" % timeit(lambda: ops.conv7_s2d_wgrad(xs, dz, dw)))
x16 = img.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
w16 = w.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
print("MIOpen conv7 fwd %.1f us" % timeit(lambda: F.conv2d(x16, w16, None, 2, 3)))
dzn = dz.permute(0, 3, 1, 2)
print("MIOpen conv7 wgrad %.1f us" % timeit(lambda: torch.ops.aten.convolution_backward(dzn, x16, w16, None, (2, 2), (3, 3), (1, 1), False, (0, 0), 1, (False, True, False))))
