#!/bin/bash
# kernel trace of a short bench run -> tools/occupancy_rounds.py table (GPU box, repo root)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 400 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/_kt2 -o run -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-roofline > gpurun_out/_kt2.log 2>&1
f=$(find gpurun_out/_kt2 -name "*kernel_trace.csv" | head -1)
head -1 $f > gpurun_out/r04_kernel_trace_header.txt
python3 tools/occupancy_rounds.py $f --after k_soft_ce 4 > gpurun_out/r04_occupancy_rounds.txt 2>&1
rm -rf gpurun_out/_kt2
cat gpurun_out/r04_kernel_trace_header.txt; cat gpurun_out/r04_occupancy_rounds.txt
