#!/usr/bin/env python3
"""event-timed 3x3 convolutions of the 128-wide stem (VOLO-D5 at 448 px: B x 224 x 224 x 128) -- csrc/conv128.hip forward / input gradient and
the quadrant weight gradient -- next to torch's (MIOpen / CK) bf16 channels_last convolution of the same shape.  usage: bench_conv128.py [B]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from autoprog_amd import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
H = W = 224
n = 3
xs = [torch.randn(B, H, W, 128, device="cuda").bfloat16() for _ in range(n)]
dys = [torch.randn(B, H, W, 128, device="cuda").bfloat16() for _ in range(n)]
w = torch.randn(128, 128, 3, 3, device="cuda") * 0.03
wf, wb = ops.conv3x3_pack(w)
dw = torch.zeros(128, 128, 3, 3, device="cuda")


def timeit(fn, it=6):
    for i in range(2):
        fn(i)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for i in range(it):
        fn(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / it


fl = 2.0 * B * H * W * 128 * 128 * 9
t = timeit(lambda i: ops.conv3x3_c64(xs[i % n], wf, True))
print("conv128 fwd + stats   %8.1f us  %6.1f TFLOP/s" % (t, fl / t / 1e6))
t = timeit(lambda i: ops.conv3x3_c64(xs[i % n], wf))
print("conv128 fwd           %8.1f us  %6.1f TFLOP/s" % (t, fl / t / 1e6))
t = timeit(lambda i: ops.conv3x3_c64(dys[i % n], wb))
print("conv128 dgrad         %8.1f us  %6.1f TFLOP/s" % (t, fl / t / 1e6))
t = timeit(lambda i: ops.conv3x3_c64_wgrad(xs[i % n], dys[i % n], dw))
print("conv128 wgrad (4 x 64) %7.1f us  %6.1f TFLOP/s" % (t, fl / t / 1e6))
xc = [v.permute(0, 3, 1, 2) for v in xs]          # NCHW views of channels_last memory
dc = [v.permute(0, 3, 1, 2) for v in dys]
w16 = w.bfloat16().contiguous(memory_format=torch.channels_last)
t = timeit(lambda i: F.conv2d(xc[i % n], w16, None, 1, 1))
print("torch (MIOpen) fwd    %8.1f us  %6.1f TFLOP/s" % (t, fl / t / 1e6))
t = timeit(lambda i: torch.ops.aten.convolution_backward(dc[i % n], xc[i % n], w16, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [True, False, False]))
print("torch (MIOpen) dgrad  %8.1f us  %6.1f TFLOP/s" % (t, fl / t / 1e6))
t = timeit(lambda i: torch.ops.aten.convolution_backward(dc[i % n], xc[i % n], w16, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [False, True, False]))
print("torch (MIOpen) wgrad  %8.1f us  %6.1f TFLOP/s" % (t, fl / t / 1e6))
