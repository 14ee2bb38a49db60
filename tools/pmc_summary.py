#!/usr/bin/env python3
"""Summarise two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, see MI355X_MICROARCH.md "HBM") into
one small table per kernel:  python tools/pmc_summary.py <fetch_counter_collection.csv> <write_...csv> <out.csv>

Units and gfx950 corrections as that guide prescribes: FETCH_SIZE / WRITE_SIZE are reported in KiB-like units of
1 KB; on gfx950 FETCH_SIZE tallies 128-byte requests at 64 bytes, so reads of wide coalesced streams are doubled
("fetch_corrected").  WRITE_SIZE is exact for 16-byte-per-lane streaming stores.  Narrow / scattered access widths are
uncalibrated: read those rows as ratios, not absolutes."""
import csv
import sys
from collections import defaultdict


def load(path, counter):
    acc = defaultdict(lambda: [0, 0.0])
    with open(path) as fh:
        for r in csv.DictReader(fh):
            if r["Counter_Name"] != counter:
                continue
            k = r["Kernel_Name"].split("(")[0]
            acc[k][0] += 1; acc[k][1] += float(r["Counter_Value"])
    return acc


def main():
    fetch, write, out = sys.argv[1:4]
    F = load(fetch, "FETCH_SIZE"); W = load(write, "WRITE_SIZE")
    rows = []
    for k in sorted(set(F) | set(W)):
        nf, f = F.get(k, [0, 0.0]); nw, w = W.get(k, [0, 0.0])
        n = max(nf, nw)
        rows.append((k, n, f * 1024 / max(nf, 1), 2 * f * 1024 / max(nf, 1), w * 1024 / max(nw, 1)))
    rows.sort(key=lambda r: -(r[3] + r[4]) * r[1])
    with open(out, "w", newline="") as fh:
        wr = csv.writer(fh)
        wr.writerow(["kernel", "dispatches", "fetch_bytes_per_launch_raw", "fetch_bytes_per_launch_corrected_x2", "write_bytes_per_launch"])
        for r in rows:
            if r[0].startswith(("void at::native", "void (anonymous namespace)", "__amd_rocclr")):
                continue            # torch kernels of the synthetic data generator and runtime copies: not the library
            wr.writerow([r[0], r[1], "%.0f" % r[2], "%.0f" % r[3], "%.0f" % r[4]])
    for r in rows[:25]:
        print("%-40s n=%5d fetch(x2) %12.3f MB  write %10.3f MB  per launch" % (r[0][:40], r[1], r[3] / 1e6, r[4] / 1e6))


if __name__ == "__main__":
    main()
