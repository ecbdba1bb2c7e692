#!/bin/bash
# HBM bytes per ap_gemm_nt launch of a bench workload from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; one counter per pass, no
# tracing) -> gpurun_out/gemm_nt_traffic_<workload>.json.  usage (GPU box, repo root): tools/collect_traffic.sh d5 ["--fp8"]
W=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/_pmcw; i=0
for grp in FETCH_SIZE WRITE_SIZE; do
  i=$((i+1))
  timeout 400 rocprofv3 --pmc $grp --output-format csv -d gpurun_out/_pmcw/p$i -o run -- python3 bench.py --workload $W "$@" --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --prewarm-s 0 --no-calibration > gpurun_out/_pmcw$i.log 2>&1
done
python3 tools/pmc_summary.py gpurun_out/_pmcw k_gemm k_mlp_fused > gpurun_out/_pmcw_counters.txt
W=$W python3 - <<'PY'
import re, json, os, sys
sys.path.insert(0, ".")
W = os.environ["W"]
txt = open("gpurun_out/_pmcw_counters.txt").read()
fetch = write = n = 0.0
for b in re.split(r"\n(?=\S)", txt):
    if b.startswith("void k_gemm_nt<") or b.startswith("void k_gemm_nt_8p<") or b.startswith("k_gemm_nt_skinny") or b.startswith("void k_mlp_fused"):
        mf = re.search(r"FETCH_SIZE\s+avg\s+([\d.]+)\s+over (\d+)", b); mw = re.search(r"WRITE_SIZE\s+avg\s+([\d.]+)", b)
        if mf and mw:
            k = int(mf.group(2)); fetch += float(mf.group(1)) * k; write += float(mw.group(1)) * k; n += k
if n:
    import bench
    json.dump({"kernel": "k_gemm_nt_8p + k_gemm_nt + k_mlp_fused2 (all instantiations)", "workload": W, "per_gpu_batch": bench.default_batch(W), "hbm_bytes_per_launch": round((2.0 * fetch + write) / n * 1024.0),
               "launches": int(n), "formula": "(2*FETCH_SIZE + WRITE_SIZE) KB per dispatch, dispatch-weighted over the k_gemm_nt* and k_mlp_fused* instantiations",
               "src_sha256": bench.kernel_source_hash()}, open("gpurun_out/gemm_nt_traffic_%s.json" % W, "w"))
    print(open("gpurun_out/gemm_nt_traffic_%s.json" % W).read())
PY
rm -rf gpurun_out/_pmcw
