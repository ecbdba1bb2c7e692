"""3x3 stem convolution kernels at the VOLO-D1 size (B = 128, 112 x 112 x 64), with and without the BatchNorm input transform."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from autoprog_amd import ops
B, H = 128, 112
xs = [torch.randn(B, H, H, 64, device="cuda").to(torch.bfloat16) for _ in range(3)]
dy = torch.randn(B, H, H, 64, device="cuda").to(torch.bfloat16)
w = torch.randn(64, 64, 3, 3, device="cuda") * 0.05
wf, wb = ops.conv3x3_pack(w)
bn = tuple(torch.rand(64, device="cuda") + 0.5 for _ in range(4))
dw = torch.zeros(64, 64, 3, 3, device="cuda")
def tm(fn, n=12):
    for i in range(3): fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n): fn(i % 3)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
print("forward + stats      plain %6.1f us   bn input %6.1f us" % (tm(lambda i: ops.conv3x3_c64(xs[i], wf, True)), tm(lambda i: ops.conv3x3_c64(xs[i], wf, True, bn_in=bn))))
print("weight gradient      plain %6.1f us   bn input %6.1f us" % (tm(lambda i: ops.conv3x3_c64_wgrad(xs[i], dy, dw)), tm(lambda i: ops.conv3x3_c64_wgrad(xs[i], dy, dw, bn_in=bn))))
print("input gradient       plain %6.1f us   + BatchNorm-backward sums %6.1f us" % (tm(lambda i: ops.conv3x3_c64(xs[i], wb)), tm(lambda i: ops.conv3x3_c64_bwd_stats(dy, wb, xs[i], bn))))
