"""Condense a rocprofv3 SQ-counter pass over the bench into one line per kernel: launches per step and, per launch,
matrix-pipe busy cycles against shader busy cycles (the utilisation figure used in profiles/README.md), plus the share of
wave time spent parked (SQ_WAIT_ANY) or issue-stalled (SQ_WAIT_INST_ANY).

Normalisation (calibrated on an 8192^3 GEMM, profiles/r01_pmc_w4_vs_vendor_8192.csv): SQ_VALU_MFMA_BUSY_CYCLES is summed
over the 1024 SIMDs (8192^3 / (32*32*16) MFMAs x 32 cycles = 1.0737e9 exactly), SQ_BUSY_CYCLES over 32 shader engines, so
matrix-pipe utilisation = (MFMA_BUSY / 1024) / (SQ_BUSY / 32). A launch held to 160 of the 256 CUs cannot exceed 0.625.

  python tools/summarize_pmc_sq.py <dir> <steps> > profiles/rNN_pmc_mfma.csv"""
import csv
import glob
import re
import sys
from collections import defaultdict


def main():
    d, steps = sys.argv[1], int(sys.argv[2])
    agg = defaultdict(lambda: defaultdict(float))
    cnt = defaultdict(set)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]).split("(")[0]
            agg[n][r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[n].add(r["Dispatch_Id"])
    out = csv.writer(sys.stdout)
    out.writerow(["kernel", "launches_per_step", "mfma_busy_Mcycles_per_launch", "sq_busy_Mcycles_per_launch",
                  "mfma_pipe_utilisation", "wait_any_over_wave_cycles", "wait_inst_over_wave_cycles",
                  "valu_inst_over_wave_cycles", "lds_inst_over_wave_cycles"])
    rows = []
    for n, c in agg.items():
        k = len(cnt[n])
        busy, mf, wc = c.get("SQ_BUSY_CYCLES", 0.0), c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), c.get("SQ_WAVE_CYCLES", 0.0)
        rows.append((busy, [n[:90], f"{k / steps:.1f}", f"{mf / k / 1e6:.3f}", f"{busy / k / 1e6:.3f}",
                            f"{mf / busy / 32:.3f}" if busy else "", f"{c.get('SQ_WAIT_ANY', 0) / wc:.3f}" if wc else "",
                            f"{c.get('SQ_WAIT_INST_ANY', 0) / wc:.3f}" if wc else "",
                            f"{c.get('SQ_ACTIVE_INST_VALU', 0) / wc:.3f}" if wc else "",
                            f"{c.get('SQ_ACTIVE_INST_LDS', 0) / wc:.3f}" if wc else ""]))
    for _, r in sorted(rows, key=lambda x: -x[0]):
        out.writerow(r)
    tb = sum(c.get("SQ_BUSY_CYCLES", 0.0) for c in agg.values())
    tm = sum(c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) for c in agg.values())
    out.writerow(["TOTAL", "", f"{tm / steps / 1e6:.1f} per step", f"{tb / steps / 1e6:.1f} per step", f"{tm / tb / 32:.3f}", "", "", "", ""])


if __name__ == "__main__":
    main()
