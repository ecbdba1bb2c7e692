"""Weight-gradient (TN) group timing on the VOLO-D1 block shapes, operands rotated over > 256 MiB: tools/bench_tn.py [reps]
Compares AP_GEMM_TN_8P=1 (csrc/gemm_tn8p.h) with =0 (128 x 128-tile kernel) -- run once per setting (the switch is read once).
Second table: the same problems of SIX transformer blocks in ONE launch (what functional's weight-gradient window issues), per block."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from autoprog_amd import ops

def group(M, dims, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    mk = lambda n: torch.randn(M, n, device="cuda", generator=g).to(torch.bfloat16)
    probs = []
    for (n1, n2, cs) in dims:
        a, b = mk(n1), mk(n2)
        w = (torch.rand(M, device="cuda", generator=g) < 0.9).to(torch.bfloat16) if cs == "w" else None
        probs.append((a, b, torch.zeros(n1, n2, device="cuda"), n1, n2, torch.zeros(n1, device="cuda") if cs else None, w, 1.1, 1.0))
    return probs

SHAPES = {
    "transformer block (C=384)": (25088, [(384, 1152, "w"), (1152, 384, "1"), (384, 384, "w"), (1152, 384, "1")]),
    "outlooker block (C=192)": (100352, [(192, 576, "1"), (576, 192, "1"), (192, 192, "1"), (192, 192, None)]),
}
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
for name, (M, dims) in SHAPES.items():
    sets = [group(M, dims, s) for s in range(3)]
    for s in sets: ops.gemm_tn_acc_grouped(s)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for r in range(reps): ops.gemm_tn_acc_grouped(sets[r % 3])
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    fl = sum(2.0 * M * n1 * n2 for n1, n2, _ in dims)
    by = sum(2.0 * M * (n1 + n2) for n1, n2, _ in dims)
    print("%-28s AP_GEMM_TN_8P=%s  %7.1f us  %6.0f TFLOP/s  %5.2f TB/s operands" % (name, os.environ.get("AP_GEMM_TN_8P", "1"), us, fl / us * 1e-6, by / us * 1e-6))

# ---- six transformer blocks in one launch: 240 tiles, no token axis cut, plain read-add-stores (operands of every block distinct)
M, dims = SHAPES["transformer block (C=384)"]
for nblk in (1, 2, 3, 6):
    sets = [sum((group(M, dims, 10 * s + b) for b in range(nblk)), []) for s in range(2)]
    for s in sets: ops.gemm_tn_acc_grouped(s)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for r in range(reps): ops.gemm_tn_acc_grouped(sets[r % 2])
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    fl = nblk * sum(2.0 * M * n1 * n2 for n1, n2, _ in dims)
    by = nblk * sum(2.0 * M * (n1 + n2) for n1, n2, _ in dims)
    print("%d transformer block(s) per launch   %7.1f us = %6.1f us per block  %6.0f TFLOP/s  %5.2f TB/s operands" % (nblk, us, us / nblk, fl / us * 1e-6, by / us * 1e-6))
