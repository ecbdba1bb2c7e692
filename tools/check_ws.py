#!/usr/bin/env python3
"""the weight-stationary K = 192 kernel (csrc/gemm_ws.h) against the 8-phase kernel on the same operands: bit-equality of outputs and codes,
and event times with rotating buffers.  Runs itself a second time with AP_GEMM_WS=0 for the reference."""
import os, sys, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from autoprog_amd import ops

def run(M, N, K=192):
    g = torch.Generator(device="cuda").manual_seed(M + N)
    res = {}
    a = torch.randn(M, K, device="cuda", generator=g).bfloat16()
    w = (torch.randn(N, K, device="cuda", generator=g) * K ** -0.5).bfloat16()
    rs = dict(row_scale=(torch.rand(M // 196 + 1, device="cuda", generator=g) > 0.2).float() / 0.8, rows_per_scale=196) if K == 384 else {}
    bias = torch.randn(N, device="cuda", generator=g) * 0.2
    codes = torch.empty(M, N, device="cuda", dtype=torch.uint8)
    res["gelu_out"] = ops.gemm_nt(a, w, bias=bias, gelu=True, preact_out=codes, preact_grad=2, **rs)
    res["gelu_codes"] = codes
    mul = torch.randint(0, 256, (M, N), device="cuda", dtype=torch.uint8, generator=g)
    res["mul8_out"] = ops.gemm_nt(a, w, mul_by=mul, **rs)
    def timeit(fn, n=30):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3
    bufs = [(torch.randn(M, K, device="cuda").bfloat16(), torch.empty(M, N, device="cuda", dtype=torch.bfloat16), torch.empty(M, N, device="cuda", dtype=torch.uint8),
             torch.randint(0, 256, (M, N), device="cuda", dtype=torch.uint8)) for _ in range(6)]
    st = {"i": 0}
    def f_gelu():
        x, o, c, _ = bufs[st["i"] % 6]; st["i"] += 1
        ops.gemm_nt(x, w, bias=bias, gelu=True, preact_out=c, preact_grad=2, out=o, **rs)
    def f_mul():
        x, o, _, c = bufs[st["i"] % 6]; st["i"] += 1
        ops.gemm_nt(x, w, mul_by=c, out=o, **rs)
    print("AP_GEMM_WS=%s  M %d N %d K %d: fc1 + GELU (table, codes) %.1f us   * codes %.1f us" % (os.environ.get("AP_GEMM_WS", "1"), M, N, K, timeit(f_gelu), timeit(f_mul)))
    return res

if __name__ == "__main__":
    shapes = [tuple(int(t) for t in v.split("x")) for v in os.environ.get("WS_SHAPES", "100352x576x192,16384x192x192,204800x576x192").split(",")]
    if os.environ.get("AP_GEMM_WS") == "0":
        out = {}
        for sh in shapes:
            for k, v in run(*sh).items(): out["%s_%s" % ("_".join(map(str, sh)), k)] = v.cpu()
        torch.save(out, "/tmp/ws_ref.pt")
    else:
        subprocess.check_call([sys.executable, __file__], env=dict(os.environ, AP_GEMM_WS="0"))
        ref = torch.load("/tmp/ws_ref.pt")
        ok = True
        for sh in shapes:
            for k, v in run(*sh).items():
                r = ref["%s_%s" % ("_".join(map(str, sh)), k)]
                same = torch.equal(v.cpu(), r)
                if not same:
                    d = (v.cpu().float() - r.float()).abs()
                    print("  MISMATCH %s: %d elements differ, max %.4g" % (k, int((d > 0).sum()), float(d.max())))
                ok &= same
        print("bit-identical to the 8-phase kernel:", ok)
