#!/usr/bin/env python3
"""stem conv3x3 (64 -> 64, NHWC bf16): HIP kernel vs torch/MIOpen -- correctness and time.  python tools/bench_conv.py [B] [H]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from autoprog_amd import ops
BF16 = torch.bfloat16
torch.backends.cudnn.benchmark = True
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
H = int(sys.argv[2]) if len(sys.argv) > 2 else 112
torch.manual_seed(0)
for (b, h, w) in [(2, 20, 37), (3, 33, 16), (B, H, H)]:
    x = torch.randn(b, h, w, 64, device="cuda").to(BF16)
    wt = torch.randn(64, 64, 3, 3, device="cuda") * 0.05
    wf, wb = ops.conv3x3_pack(wt)
    y, st = ops.conv3x3_c64(x, wf, True)
    stats = st.double().sum(0)
    ref = F.conv2d(x.float().permute(0, 3, 1, 2), wt.to(BF16).float(), None, 1, 1).permute(0, 2, 3, 1)
    print("fwd  B%d %dx%d rel err %.2e | stats sum err %.2e sq err %.2e" % (b, h, w, rel(y, ref), rel(stats[0], y.float().sum((0, 1, 2))), rel(stats[1], y.float().pow(2).sum((0, 1, 2)))))
    dy = torch.randn_like(x)
    dx = ops.conv3x3_c64(dy, wb)
    refdx = F.conv_transpose2d(dy.float().permute(0, 3, 1, 2), wt.to(BF16).float(), None, 1, 1).permute(0, 2, 3, 1)
    print("dgrad rel err %.2e" % rel(dx, refdx))
x = torch.randn(B, H, H, 64, device="cuda").to(BF16)
xn = x.permute(0, 3, 1, 2)      # NCHW view of channels_last memory
wt16 = wt.to(BF16).contiguous(memory_format=torch.channels_last)
flops = 2.0 * B * H * H * 64 * 576
t = timeit(lambda: ops.conv3x3_c64(x, wf))
print("HIP    conv3x3 fwd  %.1f us  %.0f TFLOP/s" % (t, flops / t / 1e6))
t = timeit(lambda: ops.conv3x3_c64(x, wf, True))
print("HIP    conv3x3 fwd+stats %.1f us" % t)
t = timeit(lambda: F.conv2d(xn, wt16, None, 1, 1))
print("MIOpen conv3x3 fwd  %.1f us  %.0f TFLOP/s" % (t, flops / t / 1e6))
# ---- weight gradient
for (b, h, w) in [(2, 20, 37), (3, 33, 16), (B, H, H)]:
    x = torch.randn(b, h, w, 64, device="cuda").to(BF16)
    dy = torch.randn(b, h, w, 64, device="cuda").to(BF16)
    dw = torch.zeros(64, 64, 3, 3, device="cuda")
    ops.conv3x3_c64_wgrad(x, dy, dw)
    ref = torch.nn.grad.conv2d_weight(x.float().permute(0, 3, 1, 2), (64, 64, 3, 3), dy.float().permute(0, 3, 1, 2), stride=1, padding=1)
    print("wgrad B%d %dx%d rel err %.2e" % (b, h, w, rel(dw, ref)))
t = timeit(lambda: ops.conv3x3_c64_wgrad(x, dy, dw))
print("HIP    conv3x3 wgrad %.1f us  %.0f TFLOP/s (incl. slab reduce)" % (t, flops / t / 1e6))
xn = x.permute(0, 3, 1, 2); dyn = dy.permute(0, 3, 1, 2)
t = timeit(lambda: torch.ops.aten.convolution_backward(dyn, xn, wt16, None, (1, 1), (1, 1), (1, 1), False, (0, 0), 1, (False, True, False)))
print("MIOpen conv3x3 wgrad %.1f us" % t)
