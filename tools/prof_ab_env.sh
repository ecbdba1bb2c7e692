#!/bin/bash
# kernel-trace A/B of one environment switch inside the training step (GPU box): tools/prof_ab_env.sh VAR "v0 v1" outprefix
VAR=$1; VALS=$2; OUT=$3
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p $(dirname gpurun_out/$OUT)
for v in $VALS; do
  export $VAR=$v
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/_prof$v -o run -- python3 bench.py --steps 7 --warmup 3 --prewarm-s 0 --no-calibration --no-cpu-baseline --no-roofline > gpurun_out/_prof$v.log 2>&1
  f=$(find gpurun_out/_prof$v -name "*kernel_trace.csv" | head -1)
  python tools/prof_summary.py $f --after k_soft_ce 6 > gpurun_out/${OUT}_$v.txt
  rm -rf gpurun_out/_prof$v
  echo "== $VAR=$v"; head -14 gpurun_out/${OUT}_$v.txt | cut -c1-70,90-150; tail -1 gpurun_out/${OUT}_$v.txt
done
