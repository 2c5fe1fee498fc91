"""Condense two rocprofv3 counter_collection CSVs (FETCH_SIZE pass, WRITE_SIZE pass - they cannot share a pass on gfx950)
into one per-kernel table: launches per step and bytes per launch at the L2's memory side.

  python tools/summarize_pmc.py <fetch_dir> <write_dir> <steps> > profiles/rNN_pmc_traffic.csv

FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half the bytes of wide coalesced reads
(MI355X_MICROARCH.md "HBM"), so the corrected column doubles it. Infinity-Cache hits are included in both counters."""
import csv
import glob
import re
import sys
from collections import defaultdict


def load(d):
    agg = defaultdict(lambda: [0, 0.0])
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]).split("(")[0]
            agg[n][0] += 1
            agg[n][1] += float(r["Counter_Value"])
    return agg


def main():
    fdir, wdir, steps = sys.argv[1], sys.argv[2], int(sys.argv[3])
    F, W = load(fdir), load(wdir)
    out = csv.writer(sys.stdout)
    out.writerow(["kernel", "launches_per_step", "fetch_raw_MB_per_launch", "fetch_corrected_MB_per_launch",
                  "write_MB_per_launch", "GB_per_step"])
    tot = 0.0
    for k, (n, v) in sorted(F.items(), key=lambda kv: -(2 * kv[1][1] + W.get(kv[0], [0, 0.0])[1])):
        wn, wv = W.get(k, [0, 0.0])
        gb = (2 * v + wv) * 1024 / steps / 1e9
        tot += gb
        out.writerow([k[:90], f"{n / steps:.1f}", f"{v * 1024 / n / 1e6:.1f}", f"{2 * v * 1024 / n / 1e6:.1f}",
                      f"{wv * 1024 / max(1, wn) / 1e6:.1f}", f"{gb:.2f}"])
    out.writerow(["TOTAL", "", "", "", "", f"{tot:.2f}"])


if __name__ == "__main__":
    main()
