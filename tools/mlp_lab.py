#!/usr/bin/env python3
"""times the lab builds of the fused MLP kernel (tools/mlp_lab/libmlp_abl*.so, `make -C tools/mlp_lab`): which part of the kernel costs what.
   python tools/mlp_lab.py [rows ...]"""
import ctypes
import glob
import os
import re
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from autoprog_amd._lib import MlpFusedArgs  # noqa: E402

NAMES = {1: "no MFMA", 2: "no fragment reads", 4: "no DMA", 16: "no hidden stores"}


def main():
    rows = [int(v) for v in sys.argv[1:]] or [1024, 25088]
    libs = sorted(glob.glob(os.path.join(ROOT, "tools", "mlp_lab", "libmlp_abl*.so")), key=lambda p: int(re.findall(r"abl(\d+)", p)[0]))
    C, H = 384, 1152
    g = torch.Generator().manual_seed(0)
    w1 = (torch.randn(H, C, generator=g) * 0.05).cuda().bfloat16()
    w2 = (torch.randn(C, H, generator=g) * 0.03).cuda().bfloat16()
    b1 = torch.randn(H, generator=g).cuda() * 0.1
    b2 = torch.randn(C, generator=g).cuda() * 0.1
    st = torch.cuda.current_stream().cuda_stream
    for M in rows:
        R = 4
        xs = [torch.randn(M, C, generator=g).cuda().bfloat16() for _ in range(R)]
        outs = [torch.empty(M, C, dtype=torch.bfloat16, device="cuda") for _ in range(R)]
        hid = [torch.empty(M, H, dtype=torch.bfloat16, device="cuda") for _ in range(R)]
        codes = [torch.randint(0, 255, (M, H), dtype=torch.uint8, device="cuda") for _ in range(R)]
        for path in libs:
            abl = int(re.findall(r"abl(\d+)", path)[0])
            lib = ctypes.CDLL(path)
            lib.ap_mlp_fused.restype = ctypes.c_int
            lib.ap_mlp_fused.argtypes = [ctypes.POINTER(MlpFusedArgs), ctypes.c_void_p]
            res = []
            for bwd in (0, 1):
                def run(i):
                    a = MlpFusedArgs()
                    a.x, a.ldx = xs[i % R].data_ptr(), C
                    a.wa, a.ldwa = (w2.t().contiguous() if False else w1).data_ptr(), C
                    a.wb, a.ldwb = w2.data_ptr(), H
                    a.out, a.ldo = outs[i % R].data_ptr(), C
                    a.hidden_out, a.ldh = hid[i % R].data_ptr(), H
                    a.codes = codes[i % R].data_ptr()
                    if not bwd:
                        a.bias1, a.bias2 = b1.data_ptr(), b2.data_ptr()
                        a.residual, a.ldr = xs[(i + 1) % R].data_ptr(), C
                    a.rows_per_scale = 1
                    a.m, a.c, a.hidden, a.backward = M, C, H, bwd
                    rc = lib.ap_mlp_fused(ctypes.byref(a), st)
                    assert rc == 0, rc
                for i in range(3):
                    run(i)
                reps = 16
                ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
                ev[0].record()
                for i in range(reps):
                    run(i)
                    ev[i + 1].record()
                torch.cuda.synchronize()
                ts = sorted(ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(reps))
                res.append(ts[len(ts) // 2])
            what = " + ".join(v for k, v in NAMES.items() if abl & k) or "the kernel"
            print("rows %6d  abl %2d  forward %7.1f us  backward %7.1f us   (%s)" % (M, abl, res[0], res[1], what), flush=True)


if __name__ == "__main__":
    main()
