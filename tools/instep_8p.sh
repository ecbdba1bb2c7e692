#!/bin/bash
# AP_GEMM_8P modes inside the training step: per-shape HIP-event table and ms/step.  usage: tools/instep_8p.sh [workload] [modes...]
W=${1:-d1}; shift
for m in ${@:-0 1 3}; do
  echo "== AP_GEMM_8P=$m $W"
  AP_GEMM_8P=$m AP_GEMM_TABLE=1 python bench.py --workload $W --no-cpu-baseline 2>&1 | grep -E "^ +[0-9]+ +[0-9]+ +[0-9]+ |ms_per_step" | sed -E 's/.*("ms_per_step": [0-9.]+).*/\1/' | head -19
done
