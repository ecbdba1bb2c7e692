import os, sys, torch
sys.path.insert(0, "/root/repo")
from autoprog_amd import ops
from autoprog_amd._lib import lib
for (M, K) in [(12544, 768), (12544, 3072), (25088, 384)]:
    xs = [torch.randn(M, K, device="cuda").to(torch.bfloat16) for _ in range(8)]
    ys = [torch.empty(M, K, dtype=torch.uint8, device="cuda") for _ in range(8)]
    one = torch.ones(1, device="cuda"); amax = torch.zeros(1, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    for am in (None, amax):
        def f(i): lib.ap_quantize_fp8(xs[i].data_ptr(), ys[i].data_ptr(), M * K, one.data_ptr(), am.data_ptr() if am is not None else None, st)
        for i in range(8): f(i)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for r in range(80): f(r % 8)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 80
        print(M, K, "amax" if am is not None else "no amax", "%.1f us  %.2f TB/s" % (us, M * K * 3 / us * 1e-6))
