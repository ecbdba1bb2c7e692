#!/usr/bin/env python3
"""Micro-benchmark of the HBM-bound kernels on VOLO-D1 shapes (B=128, 224 px): us and achieved GB/s of
ALGORITHMIC bytes.  Run on the GPU box: python tools/bench_misc.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from autoprog_amd import ops

def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3

def row(name, us, nbytes):
    print("%-26s %9.1f us %9.1f GB/s" % (name, us, nbytes / us / 1e3))

B = 128
dev = "cuda"
for T, C in [(B * 784, 192), (B * 196, 384)]:
    x = torch.randn(T, C, device=dev).bfloat16(); g = torch.ones(C, device=dev); b = torch.zeros(C, device=dev)
    y, m, r = ops.layernorm_fwd(x, g, b, 1e-5)
    row("ln_fwd %dx%d" % (T, C), timeit(lambda: ops.layernorm_fwd(x, g, b, 1e-5)), 2 * T * C * 2)
    dy = torch.randn_like(x); dg = torch.zeros(C, device=dev); db = torch.zeros(C, device=dev)
    row("ln_bwd(+res) %dx%d" % (T, C), timeit(lambda: ops.layernorm_bwd(dy, x, g, m, r, dy, dg, db)), 4 * T * C * 2)
N, H = 196, 12
qkv = torch.randn(B * N, 3 * 384, device=dev).bfloat16()
o, lse = ops.mhsa_fwd(qkv, B, N, H, 32 ** -0.5)
row("mhsa_fwd", timeit(lambda: ops.mhsa_fwd(qkv, B, N, H, 32 ** -0.5)), 4 * B * N * 384 * 2)
do = torch.randn_like(o)
row("mhsa_bwd", timeit(lambda: ops.mhsa_bwd(qkv, o, do, lse, B, N, H, 32 ** -0.5)), (3 + 1 + 1 + 3) * B * N * 384 * 2)
v = torch.randn(B, 28, 28, 192, device=dev).bfloat16(); lg = torch.randn(B * 196, 488, device=dev).bfloat16()
row("outlook_fwd", timeit(lambda: ops.outlook_fwd(v, lg, 6, 32 ** -0.5)), (2 * v.numel() + B * 196 * 486) * 2)
dyo = torch.randn_like(v)
row("outlook_bwd", timeit(lambda: ops.outlook_bwd(v, lg, dyo, 6, 32 ** -0.5)), (3 * v.numel() + 2 * B * 196 * 486) * 2 + v.numel() * 2)
lgt = torch.randn(B * N, 1000, device=dev).bfloat16(); tgt = torch.rand(B, 1000, 2 + N, device=dev)
row("soft_ce", timeit(lambda: ops.soft_ce_fwd_bwd(lgt, 1000, tgt[:, :, 2:], tgt.stride(0), tgt.stride(1), tgt.stride(2), N, 1e-4)), B * N * 1000 * 8)
xp = torch.randn(B, 28, 28, 192, device=dev).bfloat16()
row("avgpool_fwd", timeit(lambda: ops.avgpool2_fwd(xp)), xp.numel() * 2 * 1.25)
row("mix_swap", timeit(lambda: ops.mix_token_swap(xp, 4, 20, 4, 20)), xp.numel() * 4)
w = torch.randn(1152, 384, device=dev)
row("cast+transpose 1152x384", timeit(lambda: (ops.cast_bf16(w), ops.cast_transpose_bf16(w))), w.numel() * 8)
