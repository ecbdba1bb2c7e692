#!/bin/bash
# compile-time ablation of k_gemm_nt_p on the GPU box: AP_ABL bits 4 = no MFMA, 16 = no DMA, 32 = no fragment reads
for abl in 0 4 16 32 20 36 48 52; do
  touch autoprog_amd/csrc/gemm.hip
  make -C autoprog_amd/csrc EXTRA="-DAP_STAMP -DAP_ABL=$abl" > /dev/null 2>&1
  AP_GEMM_STAMPS=1 AP_GEMM_DBG=1 AP_GEMM_NT_P=1 timeout 120 python tools/sweep_nt.py ${1:-65536x1152x384} 2>&1 | grep "stamps" | sed "s/^/abl$abl /"
done
