"""Grouped weight-gradient launch of one transformer / outlooker block, a few times: target of the FETCH_SIZE passes that compare
AP_GEMM_TN_PLACE=0/1 (rocprofv3 --pmc FETCH_SIZE -- python3 tools/tn_place_probe.py)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from autoprog_amd import ops

def main():
    dev = torch.device("cuda:0")
    T2, T1, P = 128 * 196, 128 * 784, 128 * 196
    groups = [[(T2, 1152, 384), (T2, 384, 384), (T2, 1152, 384), (T2, 384, 1152)],
              [(T1, 192, 192), (P, 486, 192), (T1, 192, 192), (T1, 576, 192), (T1, 192, 576)]]
    for shapes in groups:
        probs = []
        for M, N1, N2 in shapes:
            a = torch.randn(M, ops.round_up(N1, 8), device=dev).bfloat16()
            b = torch.randn(M, ops.round_up(N2, 8), device=dev).bfloat16()
            probs.append((a, b, torch.zeros(N1, N2, device=dev), N1, N2, torch.zeros(N1, device=dev)))
        for _ in range(10):
            ops.gemm_tn_acc_grouped(probs)
        torch.cuda.synchronize()

if __name__ == "__main__":
    main()
