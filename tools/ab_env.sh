#!/bin/bash
# A/B of one environment switch inside the training step, interleaved runs on ONE box.  usage: tools/ab_env.sh VAR "v0 v1 ..." [rounds] [bench args...]
VAR=$1; VALS=$2; ROUNDS=${3:-2}; shift 3
for r in $(seq 1 $ROUNDS); do
  for v in $VALS; do
    out=$(env $VAR=$v python bench.py --no-cpu-baseline "$@" 2>&1 | grep -oE '"ms_per_step": [0-9.]+')
    echo "$VAR=$v round $r $out"
  done
done
