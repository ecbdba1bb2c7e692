#!/bin/bash
# end-of-round measurement set (GPU box, from the repo root): full GPU suite, default bench line + in-step GEMM table, the other workloads
R=${ROUND:-r04}
python -m pytest tests -m gpu -x -q 2>&1 | tail -3 > gpurun_out/${R}_gputests.txt
AP_GEMM_TABLE=1 python bench.py > gpurun_out/${R}_bench_n1.json 2> gpurun_out/${R}_gemm_instep_d1.txt
python bench.py --workload stages --cpu-seconds 8 2>/dev/null | tail -1 > gpurun_out/${R}_bench_stages.json
python bench.py --workload deit_base --cpu-seconds 8 2>/dev/null | tail -1 > gpurun_out/${R}_bench_deit_base.json
python bench.py --workload d5 --cpu-seconds 8 2>/dev/null | tail -1 > gpurun_out/${R}_bench_d5_448.json
python bench.py --workload d5 --fp8 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/${R}_bench_d5_fp8.json
python tools/bench_outlook.py > gpurun_out/${R}_bench_outlook.txt 2>&1
AP_OUTLOOK_P=0 python tools/bench_outlook.py >> gpurun_out/${R}_bench_outlook.txt 2>&1
tools/ab_env.sh AP_OUTLOOK_P "0 1" 2 --no-roofline > gpurun_out/${R}_ab_outlook.txt 2>&1
cat gpurun_out/${R}_gputests.txt; head -c 600 gpurun_out/${R}_bench_n1.json; echo; for w in stages deit_base d5_448 d5_fp8; do python - <<PY
import json
d=json.load(open("gpurun_out/${R}_bench_$w.json"))
print("$w", d["value"], d["ms_per_step"], (d.get("cpu_baseline") or {}).get("value"), (d.get("roofline") or {}).get("fp8_launches_per_step"))
PY
done
cat gpurun_out/${R}_bench_outlook.txt gpurun_out/${R}_ab_outlook.txt | grep -v amdgpu
