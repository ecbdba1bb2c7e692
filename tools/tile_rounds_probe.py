#!/usr/bin/env python3
"""What does a persistent NT-GEMM launch of fewer tiles than CUs pay for its idle CUs?  (Weak 5.)

The N = 384 products of a VOLO-D1 transformer block (proj, fc2, the input gradients of qkv / fc1) are 98 x 2 = 196 tiles of 256 x 192 on
256 CUs.  Two readings were on file: "the K loop runs at the rate the CHIP delivers operand bytes" (then 60 idle CUs cost nothing, and a
launch of 256 tiles takes 256 / 196 of the time) and "tile quantisation explains a quarter of the distance from the roof" (then a launch
of 256 tiles takes the SAME time as one of 196, and a tile shape that fills the chip with the same rows would be ~20 % faster).  This
probe times the same kernel, same N and K, cold operands (rotating buffers > 600 MB), at row counts that give 128 / 196 / 224 / 256 / 258
tiles: the ratio T(256 tiles) / T(196 tiles) decides (1.0: per-tile time; 1.31: chip-wide delivery)."""
import os
import sys
os.environ.setdefault("AP_GEMM_BM224", "0")          # the question is asked of the 256-row tiles (read once per process by the library)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from autoprog_amd import ops

dev = "cuda"


def timeit(fns, iters=40):
    for f in fns:
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters):
        fns[i % len(fns)]()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    print("%-22s %6s %6s %9s %9s %9s" % ("N x K, epilogue", "rows", "tiles", "us", "us/tile*", "vs 196"))
    for N, K, epi in [(384, 384, "none"), (384, 384, "res"), (384, 1152, "none"), (384, 1152, "res"), (1152, 384, "none")]:
        base = None
        tn = N // 192 if N % 192 == 0 and N < 1024 else (N + 255) // 256
        for tiles_m in (64, 98, 112, 128, 129):
            M = tiles_m * 256
            w = (torch.randn(N, K, device=dev) * K ** -0.5).bfloat16()
            nbytes = 2.0 * (M * K + M * N) * (2 if epi == "res" else 1)
            n = min(24, max(2, int(600e6 // nbytes) + 1))
            fns = []
            for _ in range(n):
                a = torch.randn(M, K, device=dev).bfloat16()
                out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
                kw = {}
                if epi == "res":
                    kw = dict(bias=torch.randn(N, device=dev), residual=torch.randn(M, N, device=dev).bfloat16(),
                              row_scale=torch.rand(M // 196 + 1, device=dev), rows_per_scale=196)
                fns.append(lambda a=a, out=out, kw=kw: ops.gemm_nt(a, w, n=N, k=K, out=out, **kw))
            us = timeit(fns)
            tiles = tiles_m * tn
            rounds = -(-tiles // 256)
            if tiles_m == 98:
                base = us
            print("%-22s %6d %6d %9.1f %9.2f %9s" % ("%d x %d, %s" % (N, K, epi), M, tiles, us, us / rounds, "" if base is None else "%.2f" % (us / base)))
            del fns
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
