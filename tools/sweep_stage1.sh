#!/bin/bash
# which tile variant (and epilogue flavour) is fastest on the outlooker-stage shapes (K or N = 192 / 576)
for t in 2 10 11 4 1; do for e in 0 1; do
  echo "tile=$t lds_epi=$e: $(AP_GEMM_LDS_EPI=$e AP_GEMM_NT_TILE=$t python tools/bench_gemm.py nt 2>/dev/null | grep -E '^out\.' | awk '{printf "%s %.1f | ", $1, $5}')"
done; done
