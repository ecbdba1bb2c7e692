#!/bin/bash
# Round-6 end-of-round measurement set (GPU box, repo root; run through gpurun in two or three calls: each block is independent).
#   tools/round_end_r06.sh tests | d1 | others | d5prof
R=r06
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
case "$1" in
tests)
  (time python -m pytest tests -m gpu -x -q) > gpurun_out/${R}_gputests.txt 2>&1; tail -4 gpurun_out/${R}_gputests.txt ;;
d1)
  ROUND=$R tools/collect_profiles.sh > gpurun_out/${R}_collect.log 2>&1
  AP_GEMM_TABLE=1 python bench.py --no-cpu-baseline > /dev/null 2> gpurun_out/${R}_gemm_instep_d1.txt
  head -c 700 gpurun_out/${R}_bench_n1.json; echo; head -30 gpurun_out/${R}_kernel_table.txt ;;
others)
  python bench.py --workload stages --cpu-seconds 8 2>/dev/null | tail -1 > gpurun_out/${R}_bench_stages.json
  python bench.py --workload stages --stage-blocks --steps 40 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/${R}_bench_stages_blocks.json
  python bench.py --workload stages --stage-blocks --steps 40 --graph --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/${R}_bench_stages_blocks_graph.json
  python bench.py --workload stages --search-mix --steps 48 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/${R}_bench_stages_search_mix.json
  python bench.py --workload stages --search-mix --graph --steps 48 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/${R}_bench_stages_search_mix_graph.json
  python bench.py --workload deit_base --cpu-seconds 8 2>/dev/null | tail -1 > gpurun_out/${R}_bench_deit_base.json
  python bench.py --workload d5 --cpu-seconds 8 2>/dev/null | tail -1 > gpurun_out/${R}_bench_d5_448.json
  python bench.py --workload d5 --fp8 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/${R}_bench_d5_fp8.json
  { for v in "volo_h12_l9 128" "volo_h12_l12 160" "volo_h12_l15 192" "volo_h12_l18 224"; do set -- $v
      e=$(python bench.py --variant $1 --res $2 --no-cpu-baseline --no-roofline --no-calibration 2>/dev/null | tail -1 | grep -oE "\"ms_per_step\": [0-9.]+")
      g=$(python bench.py --variant $1 --res $2 --graph --no-cpu-baseline --no-roofline --no-calibration 2>/dev/null | tail -1 | grep -oE "\"ms_per_step\": [0-9.]+")
      echo "$1 $2 px: eager $e | graph $g"; done; } > gpurun_out/${R}_exp_graph.txt
  timeout 600 python tools/soak.py 300 2>&1 | grep -v amdgpu | tail -16 > gpurun_out/${R}_soak_300.txt
  for w in stages stages_blocks stages_blocks_graph stages_search_mix stages_search_mix_graph deit_base d5_448 d5_fp8; do python - <<PY
import json
d=json.load(open("gpurun_out/${R}_bench_$w.json"))
print("$w", d["value"], d["ms_per_step"], (d.get("roofline") or {}).get("frac"), (d.get("roofline") or {}).get("fp8_launches_per_step"))
PY
  done; cat gpurun_out/${R}_exp_graph.txt; tail -4 gpurun_out/${R}_soak_300.txt ;;
d5prof)
  for f in "" "--fp8"; do
    tag=d5_bf16; [ -n "$f" ] && tag=d5_fp8
    timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/_kt -o run -- python3 bench.py --workload d5 $f --steps 4 --warmup 3 --no-cpu-baseline --no-roofline --prewarm-s 0 --no-calibration > gpurun_out/_kt.log 2>&1
    k=$(find gpurun_out/_kt -name "*kernel_trace.csv" | head -1)
    { echo "# rocprofv3 --kernel-trace --stats -- python3 bench.py --workload d5 $f --steps 4 --warmup 3 --no-cpu-baseline --no-roofline --prewarm-s 0 --no-calibration ; dispatches after the 6th k_soft_ce"; python3 tools/prof_summary.py $k --after k_soft_ce 6; } > gpurun_out/${R}_${tag}_kernel_stats.txt
    rm -rf gpurun_out/_kt
    head -24 gpurun_out/${R}_${tag}_kernel_stats.txt | cut -c1-70,96-140
  done
  python tools/bench_conv128.py > gpurun_out/${R}_bench_conv128.txt 2>&1; grep -v amdgpu gpurun_out/${R}_bench_conv128.txt
  python tools/tile_rounds_probe.py > gpurun_out/${R}_tile_rounds_probe.txt 2>&1 ;;
esac
