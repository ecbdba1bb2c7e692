#!/usr/bin/env python3
"""event-timed LayerNorm backward through the raw C ABI (pre-allocated buffers: no allocator / Python time in the loop)"""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from autoprog_amd import ops
from autoprog_amd._lib import lib
for T, C in [(25088, 384), (100352, 192)]:
    x = torch.randn(T, C, device="cuda").bfloat16(); dy = torch.randn(T, C, device="cuda").bfloat16()
    g = torch.ones(C, device="cuda"); b = torch.zeros(C, device="cuda")
    y, m, r = ops.layernorm_fwd(x, g, b, 1e-5)
    dx = torch.empty_like(x); dg = torch.zeros(C, device="cuda"); db = torch.zeros(C, device="cuda")
    nws = lib.ap_layernorm_bwd_workspace(T, C); ws = torch.empty(nws, dtype=torch.uint8, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    def bwd():
        lib.ap_layernorm_bwd(dy.data_ptr(), x.data_ptr(), g.data_ptr(), m.data_ptr(), r.data_ptr(), dy.data_ptr(), dx.data_ptr(), dg.data_ptr(), db.data_ptr(), T, C, ws.data_ptr(), nws, st)
    def fwd():
        lib.ap_layernorm_fwd(x.data_ptr(), g.data_ptr(), b.data_ptr(), y.data_ptr(), m.data_ptr(), r.data_ptr(), T, C, ctypes.c_float(1e-5), st)
    res = []
    for fn in (fwd, bwd):
        for _ in range(5): fn()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(50): fn()
        e1.record(); torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) * 1e3 / 50)
    print("grid=%s  %dx%d  fwd %.1f us (%.0f GB/s)  bwd+reduce %.1f us (%.0f GB/s)" % (os.environ.get("AP_LN_BWD_GRID", "-"), T, C, res[0], 4.0 * T * C / res[0] / 1e3, res[1], 8.0 * T * C / res[1] / 1e3))
