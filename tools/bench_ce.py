"""token-label CE kernels at the D1 size (B = 128, N = 196, 1000 classes): dense target vs label maps"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from autoprog_amd.loss import TokenLabelCrossEntropy
B, N, C = 128, 196, 1000
g = torch.Generator().manual_seed(0)
loss_fn = TokenLabelCrossEntropy(dense_weight=0.5, cls_weight=1.0, classes=C)
xc = torch.randn(B, C, device="cuda").to(torch.bfloat16)
xa = torch.randn(B, N, C, device="cuda").to(torch.bfloat16)
for name, sparse in (("dense", False), ("label maps", True)):
    t = bench.make_target(B, C, N, "cuda", g, sparse=sparse)
    f = lambda: loss_fn((xc, xa, (1, 2, 5, 6)), t)
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    print("%-12s %.1f us per loss forward (3 launches + glue)" % (name, e0.elapsed_time(e1) * 50))
