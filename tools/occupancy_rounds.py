#!/usr/bin/env python3
"""For every kernel of a rocprofv3 --kernel-trace CSV: workgroups per launch against the workgroups a chip of 256 CUs holds at once
(limited by VGPRs -- 512 per SIMD lane --, LDS -- 160 KB per CU -- and 32 waves per CU): a launch of 1.5 'rounds' runs its load ->
compute -> store chain twice for half a chip's worth of work (round 4: k_ln_fwd, 6.1 workgroups per CU where 4 fit).
   python tools/occupancy_rounds.py <kernel_trace.csv> [--after KERNEL_SUBSTR N]"""
import csv, sys
from collections import defaultdict
path = sys.argv[1]
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = 0
if "--after" in sys.argv:
    i = sys.argv.index("--after"); sub, nth = sys.argv[i + 1], int(sys.argv[i + 2]); seen = 0
    for r in rows:
        if sub in r["Kernel_Name"]:
            seen += 1
            if seen == nth:
                t0 = int(r["End_Timestamp"]); break
# kernels that ask for dynamic LDS (bytes per workgroup at the VOLO-D1 shapes; csrc/*.hip)
DYNAMIC_LDS = [("k_gemm_nt_8p", 163840), ("k_gemm_tn_8p", 150528), ("k_mhsa_fwd_p", 57344), ("k_mhsa_bwd_ds", 163840), ("k_outlook_p<512, 3, 1, true>", 75000),
               ("k_outlook_p<512, 3, 1, false>", 50000), ("k_conv3x3_c64", 163840), ("k_conv7_s2d", 40000), ("k_ln_fwd", 3072), ("k_ln_bwd", 12288)]


def col(r, *names):
    for n in names:
        if n in r and r[n] != "":
            return int(float(r[n]))
    return 0
agg = defaultdict(lambda: [0, 0.0, None])
for r in rows:
    if int(r["Start_Timestamp"]) < t0:
        continue
    wg = max(1, col(r, "Workgroup_Size_X", "Workgroup_Size") * max(1, col(r, "Workgroup_Size_Y")) * max(1, col(r, "Workgroup_Size_Z")))
    grid = max(1, col(r, "Grid_Size_X", "Grid_Size") * max(1, col(r, "Grid_Size_Y")) * max(1, col(r, "Grid_Size_Z")))
    nwg = grid // wg
    # rocprofv3 reports the allocation per SIMD32 half of a wave64: twice that is what .vgpr_count of the code object says
    vg = 2 * (col(r, "VGPR_Count", "Arch_VGPR_Count") + col(r, "Accum_VGPR_Count"))
    lds = col(r, "LDS_Block_Size", "LDS_Block_Size_v")          # static LDS only: the dynamic allocations are listed below
    for sub, dyn in DYNAMIC_LDS:
        if sub in r["Kernel_Name"]:
            lds = max(lds, dyn)
    waves = (wg + 63) // 64
    per_simd = max(1, min(8, 512 // max(vg, 1))) if vg else 8
    by_vgpr = (per_simd * 4) // waves if waves <= per_simd * 4 else 0
    by_lds = (160 * 1024) // lds if lds else 99
    by_waves = 32 // waves
    res = max(1, min(by_vgpr if by_vgpr else 1, by_lds, by_waves))
    key = (r["Kernel_Name"].split("(")[0][:70], nwg, wg, vg, lds)
    a = agg[key]; a[0] += 1; a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3; a[2] = res
print("%-70s %6s %7s %5s %5s %7s %6s %7s %9s" % ("kernel", "calls", "wgs", "thr", "vgpr", "lds", "wg/CU", "rounds", "avg_us"))
for k, (n, t, res) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:60]:
    print("%-70s %6d %7d %5d %5d %7d %6d %7.2f %9.1f" % (k[0], n, k[1], k[2], k[3], k[4], res, k[1] / (res * 256.0), t / n))
