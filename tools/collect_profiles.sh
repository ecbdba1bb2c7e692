#!/bin/bash
# Regenerates the judged measurement artifacts on the GPU box (run through gpurun from the repo root):
#   gpurun_out/${ROUND}_bench_n1.json      the default bench line (roofline + cpu_baseline)
#   gpurun_out/${ROUND}_kernel_stats.txt   rocprofv3 --kernel-trace --stats summary of the same command (timed steps only)
#   gpurun_out/${ROUND}_pmc_counters.txt   rocprofv3 --pmc passes (one counter group per pass, no tracing) per kernel
#   gpurun_out/${ROUND}_kernel_table.txt   one line per kernel: HBM bytes, TB/s, MFMA busy %, VALU busy %, LDS conflict share (tools/pmc_table.py)
#   gpurun_out/gemm_nt_traffic.json   HBM bytes per k_gemm_nt launch from FETCH_SIZE/WRITE_SIZE (gfx950 correction applied)
# Copy them into profiles/ afterwards.
ROUND=${ROUND:-r06}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
if [ -z "$PMC_ONLY" ]; then
python3 bench.py 2> gpurun_out/${ROUND}_bench.err | tail -1 > gpurun_out/${ROUND}_bench_n1.json
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/_kt -o run -- python3 bench.py --steps 7 --warmup 3 --no-cpu-baseline --no-roofline --prewarm-s 0 --no-calibration > gpurun_out/_kt.log 2>&1
f=$(find gpurun_out/_kt -name "*kernel_trace.csv" | head -1)
{ echo "# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 7 --warmup 3 --no-cpu-baseline --no-roofline --prewarm-s 0 --no-calibration ; dispatches after the 6th k_soft_ce (3 warm-up steps dropped; kept: the 7 timed steps + the 6 forward+backward-only probe steps behind fwd_loss_bwd_only_ms_per_step)"; python3 tools/prof_summary.py $f --after k_soft_ce 6; } > gpurun_out/${ROUND}_kernel_stats.txt
rm -rf gpurun_out/_kt
fi
i=0
# FETCH_SIZE and WRITE_SIZE are derived metrics that do not fit one pass together ("exceeds the capabilities of the hardware",
# after which rocprofv3 aborts and hangs): one pass each, every pass under a hard timeout
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY" "SQ_ACTIVE_INST_VALU SQ_INSTS_VALU"; do
  i=$((i+1))
  timeout 240 rocprofv3 --pmc $grp --output-format csv -d gpurun_out/_pmc/p$i -o run -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --prewarm-s 0 --no-calibration > gpurun_out/_pmc$i.log 2>&1
done
{ echo "# rocprofv3 --pmc <one group per pass> -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --prewarm-s 0 --no-calibration ; per-dispatch averages"
  echo "# FETCH_SIZE/WRITE_SIZE are in KB; gfx950 FETCH_SIZE under-reports wide coalesced reads by 2x (MI355X_MICROARCH.md): double it."
  python3 tools/pmc_summary.py gpurun_out/_pmc k_gemm k_mlp_fused k_ln k_mhsa k_outlook k_soft_ce k_bn k_adamw k_conv3x3 igemm kernel_grouped_conv; } > gpurun_out/${ROUND}_pmc_counters.txt
ROUND=$ROUND python3 - <<'PY'
import re, json, os, sys
sys.path.insert(0, ".")
R = os.environ["ROUND"]
txt = open("gpurun_out/%s_pmc_counters.txt" % R).read()
blocks = re.split(r"\n(?=\S)", txt)
fetch = write = n = 0.0
for b in blocks:
    if b.startswith("void k_gemm_nt<") or b.startswith("void k_gemm_nt_8p<") or b.startswith("void k_gemm_nt_ws<") or b.startswith("k_gemm_nt_skinny") or b.startswith("void k_mlp_fused"):
        mf = re.search(r"FETCH_SIZE\s+avg\s+([\d.]+)\s+over (\d+)", b); mw = re.search(r"WRITE_SIZE\s+avg\s+([\d.]+)", b)
        if mf and mw:
            k = int(mf.group(2)); fetch += float(mf.group(1)) * k; write += float(mw.group(1)) * k; n += k
if n:
    per = (2.0 * fetch + write) / n * 1024.0
    import bench
    json.dump({"kernel": "k_gemm_nt_8p + k_gemm_nt_ws + k_gemm_nt + k_mlp_fused2 (all instantiations)", "hbm_bytes_per_launch": round(per), "launches": int(n),
               "formula": "(2*FETCH_SIZE + WRITE_SIZE) KB per dispatch, dispatch-weighted over the k_gemm_nt* and k_mlp_fused* instantiations",
               "source": "profiles/%s_pmc_counters.txt" % R, "src_sha256": bench.kernel_source_hash()}, open("gpurun_out/gemm_nt_traffic.json", "w"))
PY
python3 tools/pmc_table.py gpurun_out/${ROUND}_pmc_counters.txt gpurun_out/${ROUND}_kernel_stats.txt > gpurun_out/${ROUND}_kernel_table.txt
rm -rf gpurun_out/_pmc
grep -A9 'k_gemm_nt_8p<1, 0>' gpurun_out/${ROUND}_pmc_counters.txt | head -12; cat gpurun_out/gemm_nt_traffic.json
