#!/usr/bin/env python3
"""event-timed LayerNorm forward / backward through the raw C ABI (pre-allocated buffers: no allocator / Python time in the loop),
inputs rotated over > 256 MiB so that nothing is served from the Infinity Cache.  AP_LIB_PATH selects another build of the library."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from autoprog_amd import ops
from autoprog_amd._lib import lib
for T, C in [(25088, 384), (100352, 384), (401408, 384), (100352, 192), (12544, 768)]:
    n = max(2, int(400e6 / (T * C * 2 * 4)))
    xs = [torch.randn(T, C, device="cuda").bfloat16() for _ in range(n)]
    dys = [torch.randn(T, C, device="cuda").bfloat16() for _ in range(n)]
    g = torch.ones(C, device="cuda"); b = torch.zeros(C, device="cuda")
    ys = [torch.empty_like(x) for x in xs]; dxs = [torch.empty_like(x) for x in xs]
    m = torch.empty(T, device="cuda"); r = torch.empty(T, device="cuda")
    dg = torch.zeros(C, device="cuda"); db = torch.zeros(C, device="cuda")
    nws = lib.ap_layernorm_bwd_workspace(T, C); ws = torch.empty(nws, dtype=torch.uint8, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    npart = ctypes.c_int(0)
    def fwd(i):
        lib.ap_layernorm_fwd(xs[i].data_ptr(), g.data_ptr(), b.data_ptr(), ys[i].data_ptr(), m.data_ptr(), r.data_ptr(), T, C, ctypes.c_float(1e-5), st)
    def bwd(i):      # the in-step form: partial rows only, the reduction rides in the weight-gradient launch
        lib.ap_layernorm_bwd_partial(dys[i].data_ptr(), xs[i].data_ptr(), g.data_ptr(), m.data_ptr(), r.data_ptr(), dys[(i + 1) % n].data_ptr(), dxs[i].data_ptr(),
                                     T, C, ws.data_ptr(), nws, ctypes.byref(npart), st)
    res = []
    for fn in (fwd, bwd):
        for i in range(n): fn(i)
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for k in range(60): fn(k % n)
        e1.record(); torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) * 1e3 / 60)
    print("grid=%s  %dx%d  fwd %.1f us (%.0f GB/s)  bwd %.1f us (%.0f GB/s)" % (os.environ.get("AP_LN_BWD_GRID", "-"), T, C, res[0], 4.0 * T * C / res[0] / 1e3, res[1], 8.0 * T * C / res[1] / 1e3))
