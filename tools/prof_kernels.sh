#!/bin/bash
# per-kernel durations of the training step under rocprofv3 (run on the GPU box): tools/prof_kernels.sh [out-name] [filter-regex]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=${1:-kstats}
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/_prof -o run -- python3 bench.py --steps 7 --warmup 3 --no-cpu-baseline --no-roofline --prewarm-s 0 --no-calibration > gpurun_out/_prof.log 2>&1
f=$(find gpurun_out/_prof -name "*kernel_trace.csv" | head -1)
python tools/prof_summary.py $f --after k_soft_ce 6 > gpurun_out/$out.txt
rm -rf gpurun_out/_prof
if [ -n "$2" ]; then grep -E "$2" gpurun_out/$out.txt | cut -c1-60,90-140; else head -40 gpurun_out/$out.txt | cut -c1-60,90-140; fi
tail -1 gpurun_out/$out.txt
