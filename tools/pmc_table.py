#!/usr/bin/env python3
"""One line per kernel from the PMC averages (tools/pmc_summary.py output) and the kernel-trace averages (tools/prof_summary.py
output):  python tools/pmc_table.py r02_pmc_counters.txt r02_kernel_stats.txt

  hbm_MB      = (2 * FETCH_SIZE + WRITE_SIZE) per launch  (FETCH_SIZE doubled: the gfx950 correction of MI355X_MICROARCH.md)
  TB/s        = hbm_MB / avg duration of the kernel-trace run (no counters attached)
  mfma_busy%  = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs * GRBM_GUI_ACTIVE / 8 XCDs)
  valu_busy%  = 4 * SQ_ACTIVE_INST_VALU / (1024 SIMDs * GRBM_GUI_ACTIVE / 8 XCDs)   (rocprof's VALUBusy: 4 cycles per wave64 VALU instruction)
  lds_confl%  = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE      (share of the LDS-array cycles that are conflict replays)
  confl/kern% = (SQ_LDS_BANK_CONFLICT / 256 CUs) / (GRBM_GUI_ACTIVE / 8)   (conflict cycles per CU over the kernel's cycles)
  wait%       = SQ_WAIT_ANY / SQ_WAVE_CYCLES               (wave time parked in s_waitcnt / barriers)"""
import re
import sys


def blocks(path):
    out, cur = {}, None
    for line in open(path):
        if line.startswith("#") or not line.strip():
            continue
        if not line.startswith(" "):
            cur = out.setdefault(line.strip(), {})
        else:
            m = re.match(r"\s+(\S+)\s+avg\s+([\d.]+)\s+over\s+(\d+)", line)
            if m and cur is not None:
                cur[m.group(1)] = (float(m.group(2)), int(m.group(3)))
    return out


def stats(path):
    out = {}
    for line in open(path):
        m = re.match(r"(.+?)\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s*$", line)
        if m and not line.startswith("kernel"):
            out[m.group(1).strip()] = (int(m.group(2)), float(m.group(4)))
    return out


def main():
    pmc, st = blocks(sys.argv[1]), stats(sys.argv[2])
    print("%-64s %7s %9s %8s %7s %10s %10s %10s %11s %6s" % ("kernel", "calls", "avg_us", "hbm_MB", "TB/s", "mfma_busy%", "valu_busy%", "lds_confl%", "confl/kern%", "wait%"))
    rows = []
    for name, c in pmc.items():
        key = next((k for k in st if k[:88] == name[:88]), None)
        calls, avg = st.get(key, (0, 0.0))
        g = lambda k: c.get(k, (0.0, 0))[0]
        hbm = (2.0 * g("FETCH_SIZE") + g("WRITE_SIZE")) * 1024.0 / 1e6
        cyc = g("GRBM_GUI_ACTIVE") / 8.0
        rows.append((calls * avg, name, calls, avg, hbm, hbm / avg if avg else 0.0,          # MB / us = TB/s
                     100.0 * g("SQ_VALU_MFMA_BUSY_CYCLES") / (1024.0 * cyc) if cyc else 0.0,
                     400.0 * g("SQ_ACTIVE_INST_VALU") / (1024.0 * cyc) if cyc else 0.0,
                     100.0 * g("SQ_LDS_BANK_CONFLICT") / g("SQ_LDS_IDX_ACTIVE") if g("SQ_LDS_IDX_ACTIVE") else 0.0,
                     100.0 * g("SQ_LDS_BANK_CONFLICT") / 256.0 / cyc if cyc else 0.0,
                     100.0 * g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES") if g("SQ_WAVE_CYCLES") else 0.0))
    for r in sorted(rows, reverse=True):
        print("%-64s %7d %9.1f %8.1f %7.2f %10.1f %10.1f %10.1f %11.1f %6.1f" % ((r[1][:64],) + r[2:]))


if __name__ == "__main__":
    main()
