#!/usr/bin/env python3
"""per-slot timeline of the fused MLP kernel (version 2) from the s_memtime stamps of a lab build (tools/mlp_lab/libmlp_abl0.so):
   python tools/mlp_stamps.py [rows] [backward]"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from autoprog_amd._lib import MlpFusedArgs  # noqa: E402


def main():
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 25088
    bwd = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    lib = ctypes.CDLL(os.path.join(ROOT, "tools", "mlp_lab", os.environ.get("MF_LAB_LIB", "libmlp_abl0.so")))
    lib.ap_mlp_fused.restype = ctypes.c_int
    lib.ap_mlp_fused.argtypes = [ctypes.POINTER(MlpFusedArgs), ctypes.c_void_p]
    C, H = 384, 1152
    g = torch.Generator().manual_seed(0)
    w1 = (torch.randn(H, C, generator=g) * 0.05).cuda().bfloat16()
    w2 = (torch.randn(C, H, generator=g) * 0.03).cuda().bfloat16()
    b1 = torch.randn(H, generator=g).cuda() * 0.1
    b2 = torch.randn(C, generator=g).cuda() * 0.1
    xs = [torch.randn(M, C, generator=g).cuda().bfloat16() for _ in range(3)]
    out = torch.empty(M, C, dtype=torch.bfloat16, device="cuda")
    hid = torch.empty(M, H, dtype=torch.bfloat16, device="cuda")
    codes = torch.randint(0, 255, (M, H), dtype=torch.uint8, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    for i in range(6):
        a = MlpFusedArgs()
        a.x, a.ldx = xs[i % 3].data_ptr(), C
        a.wa, a.ldwa = w1.data_ptr(), C
        a.wb, a.ldwb = w2.data_ptr(), H
        a.out, a.ldo = out.data_ptr(), C
        a.hidden_out, a.ldh = hid.data_ptr(), H
        a.codes = codes.data_ptr()
        if not bwd:
            a.bias1, a.bias2 = b1.data_ptr(), b2.data_ptr()
            a.residual, a.ldr = xs[(i + 1) % 3].data_ptr(), C
        a.rows_per_scale = 1
        a.m, a.c, a.hidden, a.backward = M, C, H, bwd
        assert lib.ap_mlp_fused(ctypes.byref(a), st) == 0
    torch.cuda.synchronize()
    n = 2 * 32 * 4 * 2
    buf = (ctypes.c_ulonglong * n)()
    assert lib.mf_lab_read_stamps(buf, n) == 0
    v = list(buf)

    def at(role, t, q, k):
        return v[((role * 32 + t) * 4 + q) * 2 + k]
    t0 = at(1, 0, 0, 0)
    print("rows %d %s: cycles (s_memtime) per slot of one workgroup -- A: producer wave 0, B: consumer wave 4; 'work' = barrier -> slot's work done, 'wait' = work done -> next barrier" % (M, "backward" if bwd else "forward"))
    for t in range(0, 20):
        line = "period %2d:" % t
        for q in range(4):
            nxt = at(0, t, q + 1, 0) if q < 3 else at(0, t + 1, 0, 0)
            line += "  q%d A %5d/%5d B %5d/%5d" % (q, at(0, t, q, 1) - at(0, t, q, 0), max(0, nxt - at(0, t, q, 1)) if nxt else 0,
                                                 at(1, t, q, 1) - at(1, t, q, 0), max(0, (at(1, t, q + 1, 0) if q < 3 else at(1, t + 1, 0, 0)) - at(1, t, q, 1)))
        if t < 19:
            line += "   | period %6d" % (at(1, t + 1, 0, 0) - at(1, t, 0, 0))
        print(line)
    print("total (first barrier -> output stored): %d cycles" % (at(1, 19, 0, 0) - t0))


if __name__ == "__main__":
    main()
