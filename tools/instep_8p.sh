#!/bin/bash
# AP_GEMM_8P modes inside the training step: per-shape HIP-event table and ms/step.  usage: tools/instep_8p.sh [workload] [modes...]
W=${1:-d1}; shift
for m in ${@:-0 1 3}; do
  echo "== AP_GEMM_8P=$m $W"
  AP_GEMM_8P=$m AP_GEMM_TABLE=1 python bench.py --workload $W --no-cpu-baseline > /tmp/instep.log 2>&1
  grep -E "^ +[0-9]+ +[0-9]+ +[0-9]+ " /tmp/instep.log | head -${ROWS:-19}
  grep -oE '"ms_per_step": [0-9.]+|"value": [0-9.]+' /tmp/instep.log
done
