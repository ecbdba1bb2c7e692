import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from autoprog_amd import ops
def t(M, N, K, iters=10):
    a = torch.randn(M, K, device="cuda").bfloat16(); w = (torch.randn(N, K, device="cuda") * K ** -0.5).bfloat16()
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    for _ in range(3): ops.gemm_nt(a, w, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): ops.gemm_nt(a, w, out=out)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / iters * 1e3
    print("M %6d N %5d K %5d  %9.1f us  %7.1f TFLOP/s" % (M, N, K, us, 2.0 * M * N * K / us / 1e6))
for shape in [(4096, 4096, 4096), (8192, 8192, 8192), (25088, 1152, 384), (25088, 1152, 1536), (25088, 1152, 6144), (25088, 4096, 384)]:
    t(*shape)
