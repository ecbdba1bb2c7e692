for d in 0 1 2 3 4 6 7; do echo "DBG=$d"; AP_GEMM_DBG=$d python tools/bench_gemm.py nt 2>&1 | grep -E "tr.qkv|out.v |tr.fc2|NT total"; done
