#!/usr/bin/env python3
"""Micro-benchmark of the C-ABI GEMMs on the VOLO-D1 shapes (B=128, 224 px): TFLOP/s and GB/s per shape.
Run on the GPU box:  python tools/bench_gemm.py [nt|tn|all]"""
import os
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from autoprog_amd import ops

T1, T2, P = 100352, 25088, 25088
NT = [  # (name, M, N, K, epilogue)
    ("out.v", T1, 192, 192, "none"), ("out.attn", P, 486, 192, "bias"), ("out.proj+res", T1, 192, 192, "res"),
    ("out.fc1+gelu", T1, 576, 192, "gelu"), ("out.fc2+res", T1, 192, 576, "res"), ("out.dfc2*dgelu", T1, 576, 192, "dgelu"),
    ("out.dfc1", T1, 192, 576, "none"), ("down", P, 384, 768, "bias"),
    ("tr.qkv", T2, 1152, 384, "none"), ("tr.proj+res", T2, 384, 384, "res"), ("tr.fc1+gelu", T2, 1152, 384, "gelu"),
    ("tr.fc2+res", T2, 384, 1152, "res"), ("tr.dfc2*dgelu", T2, 1152, 384, "dgelu"), ("tr.dfc1", T2, 384, 1152, "none"),
    ("tr.dqkv", T2, 384, 1152, "none"), ("aux_head", T2, 1000, 384, "bias"), ("d_aux_head", T2, 384, 1000, "none"),
]
TN = [("out.v/proj", T1, 192, 192), ("out.attn", P, 486, 192), ("out.fc1", T1, 576, 192), ("out.fc2", T1, 192, 576),
      ("down", P, 384, 768), ("tr.qkv", T2, 1152, 384), ("tr.proj", T2, 384, 384), ("tr.fc1", T2, 1152, 384),
      ("tr.fc2", T2, 384, 1152), ("aux_head", T2, 1000, 384)]


COLD = os.environ.get("AP_BENCH_COLD", "1") != "0"     # rotate over > 600 MB of operand copies (Infinity Cache is 256 MiB);
                                                         # AP_BENCH_COLD=0 reuses ONE buffer set (cache-warm: flatters kernels that
                                                         # stream more bytes -- tile choices made that way lost in the real step)


def rotating(make, nbytes):
    """make() -> (fn taking no args); returns a function cycling over enough independent buffer sets"""
    n = max(1, int(600e6 // max(nbytes, 1)) + 1) if COLD else 1
    fns = [make() for _ in range(min(n, 24))]
    state = {"i": 0}

    def call():
        fns[state["i"] % len(fns)]()
        state["i"] += 1
    return call


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3     # us


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    dev = "cuda"
    tot_t = tot_f = 0.0
    if which in ("nt", "all"):
        print("%-16s %7s %5s %5s %9s %9s %9s" % ("gemm_nt", "M", "N", "K", "us", "TFLOP/s", "GB/s"))
        for name, M, N, K, epi in NT:
            ld = ops.round_up(N, 8)
            kp = ops.round_up(K, 8)
            w = (torch.randn(N, kp, device=dev) * K ** -0.5).bfloat16()
            bias = torch.randn(N, device=dev)
            nbytes = 2.0 * (M * K + N * K + M * N) + (2.0 * M * N if epi in ("gelu", "res", "dgelu") else 0.0)

            def make():
                a = torch.randn(M, kp, device=dev).bfloat16()
                out = torch.empty(M, ld, device=dev, dtype=torch.bfloat16)
                kw = {}
                if epi in ("bias", "gelu", "res"):
                    kw["bias"] = bias
                if epi == "gelu":
                    kw["gelu"] = True
                    kw["preact_out"] = torch.empty(M, ld, device=dev, dtype=torch.bfloat16)
                if epi == "res":
                    kw["residual"] = torch.randn(M, ld, device=dev).bfloat16()
                    kw["row_scale"] = torch.rand(M // 196 + 1, device=dev)
                    kw["rows_per_scale"] = 196
                if epi == "dgelu":
                    kw["dgelu_of"] = torch.randn(M, ld, device=dev).bfloat16()
                return lambda: ops.gemm_nt(a, w, n=N, k=kp, out=out, **kw)
            us = timeit(rotating(make, nbytes))
            fl = 2.0 * M * N * K
            tot_t += us; tot_f += fl
            print("%-16s %7d %5d %5d %9.1f %9.1f %9.1f" % (name, M, N, K, us, fl / us / 1e6, nbytes / us / 1e3))
        print("NT total %.1f us, %.1f TFLOP/s" % (tot_t, tot_f / tot_t / 1e6))
    tot_t = tot_f = 0.0
    if which in ("tn", "all"):
        print("%-16s %7s %5s %5s %9s %9s %9s" % ("gemm_tn_acc", "M", "N1", "N2", "us", "TFLOP/s", "GB/s"))
        for name, M, N1, N2 in TN:
            a = torch.randn(M, ops.round_up(N1, 8), device=dev).bfloat16()
            b = torch.randn(M, ops.round_up(N2, 8), device=dev).bfloat16()
            c = torch.zeros(N1, N2, device=dev)
            us = timeit(lambda: ops.gemm_tn_acc(a, b, c))
            fl = 2.0 * M * N1 * N2
            tot_t += us; tot_f += fl
            print("%-16s %7d %5d %5d %9.1f %9.1f %9.1f" % (name, M, N1, N2, us, fl / us / 1e6, 2.0 * M * (N1 + N2) / us / 1e3))
        print("TN total %.1f us, %.1f TFLOP/s" % (tot_t, tot_f / tot_t / 1e6))
        # grouped launches: the weight gradients of one block in ONE launch (functional.wgrad_batch)
        groups = {"transformer block (qkv, proj, fc1, fc2)": [(T2, 1152, 384), (T2, 384, 384), (T2, 1152, 384), (T2, 384, 1152)],
                  "outlooker block (v, attn, proj, fc1, fc2)": [(T1, 192, 192), (P, 486, 192), (T1, 192, 192), (T1, 576, 192), (T1, 192, 576)]}
        for gname, shapes in groups.items():
            probs = []
            fl = 0.0
            for M, N1, N2 in shapes:
                a = torch.randn(M, ops.round_up(N1, 8), device=dev).bfloat16()
                b = torch.randn(M, ops.round_up(N2, 8), device=dev).bfloat16()
                probs.append((a, b, torch.zeros(N1, N2, device=dev), N1, N2, torch.zeros(N1, device=dev)))
                fl += 2.0 * M * N1 * N2
            us_g = timeit(lambda: ops.gemm_tn_acc_grouped(probs))
            us_s = timeit(lambda: [ops.gemm_tn_acc(a, b, c, n1=n1, n2=n2, colsum=cs) for a, b, c, n1, n2, cs in probs])
            print("%-44s grouped %7.1f us (%.0f TFLOP/s)   one by one %7.1f us" % (gname, us_g, fl / us_g / 1e6, us_s))


if __name__ == "__main__":
    main()
