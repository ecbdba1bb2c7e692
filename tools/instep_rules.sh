#!/bin/bash
# time candidate tile variants for given (N,K) INSIDE the training step: tools/instep_rules.sh "RULES" ["RULES" ...]
for r in "$@"; do
  echo "== AP_GEMM_NT_RULE=$r"
  AP_GEMM_NT_RULE="$r" AP_GEMM_TABLE=1 python bench.py --no-cpu-baseline 2>&1 | grep -E "^ +(100352|25088) +[0-9]+ +[0-9]+ |ms_per_step" | sed -E 's/.*("ms_per_step": [0-9.]+).*/\1/' | head -17
done
