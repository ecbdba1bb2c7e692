#!/bin/bash
# extra end-of-round evidence (GPU box, repo root): kernel stats of the d5 / stages workloads, batch scaling of the D1 step, the D1-sized
# late-state comparison after 600 steps (the synthetic batch starts to be fitted around step 300)
R=${ROUND:-r04}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for w in d5 stages; do
  timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/_ktw -o run -- python3 bench.py --workload $w --steps 4 --warmup 3 --no-cpu-baseline --no-roofline > gpurun_out/_ktw.log 2>&1
  f=$(find gpurun_out/_ktw -name "*kernel_trace.csv" | head -1)
  { echo "# rocprofv3 --kernel-trace --stats -- python3 bench.py --workload $w --steps 4 --warmup 3 --no-cpu-baseline --no-roofline ; dispatches after the 6th k_soft_ce"; python3 tools/prof_summary.py $f --after k_soft_ce 6; } > gpurun_out/${R}_${w}_kernel_stats.txt
  rm -rf gpurun_out/_ktw
done
for b in 64 128 256; do python3 bench.py --batch $b --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('batch', d['config']['per_gpu_batch'], d['value'], 'images/s', d['ms_per_step'], 'ms')"; done > gpurun_out/${R}_batch_scaling.txt
python3 tools/late_state_parity.py 600 > gpurun_out/${R}_late_state_parity_600.txt 2>&1
cat gpurun_out/${R}_batch_scaling.txt; grep -E "trained|logits|one vector|median" gpurun_out/${R}_late_state_parity_600.txt; head -12 gpurun_out/${R}_d5_kernel_stats.txt | cut -c1-150
