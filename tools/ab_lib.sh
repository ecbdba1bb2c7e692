#!/bin/bash
# A/B of library builds inside the training step on ONE box: tools/ab_lib.sh "libA libB ..." [rounds] [grep pattern of the in-step GEMM table]   ("-" = the product library)
LIBS=$1; ROUNDS=${2:-2}; PAT=${3:-"486|1000"}
for r in $(seq 1 $ROUNDS); do
  for l in $LIBS; do
    p=$l; [ "$l" = "-" ] && p=""
    AP_LIB_PATH=$p AP_GEMM_TABLE=1 python bench.py --no-cpu-baseline > /tmp/_ab.json 2> /tmp/_ab.txt
    echo "lib=$l round $r $(grep -oE '"ms_per_step": [0-9.]+' /tmp/_ab.json)"; grep -E "$PAT" /tmp/_ab.txt | head -8
  done
done
