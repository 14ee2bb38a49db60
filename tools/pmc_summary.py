#!/usr/bin/env python3
"""Summarise two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, see MI355X_MICROARCH.md "HBM") into
one small table per kernel:  python tools/pmc_summary.py <fetch_counter_collection.csv> <write_...csv> <out.csv>

Units and gfx950 corrections as that guide prescribes: FETCH_SIZE / WRITE_SIZE are reported in KiB-like units of
1 KB; on gfx950 FETCH_SIZE tallies the 128-byte requests of a wide coalesced stream (16 bytes per lane) at 64 bytes, so the
reads of such kernels -- and ONLY those, listed in STREAM16 below -- are doubled.  WRITE_SIZE is exact for 16-byte-per-lane
streaming stores.  Every other access width is uncalibrated (MI355X_MICROARCH.md "HBM"): those rows carry the raw counter
as `traffic` and calibrated = 0: read them as ratios between variants, or as an upper bound after doubling, not as absolutes."""
import csv
import sys
from collections import defaultdict


# kernels whose global reads are 16-byte-per-lane coalesced streams (the x2 correction is documented for exactly that)
STREAM16 = {"row_reduce_kernel", "normalize_kernel"}


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
        raw = f * 1024 / max(nf, 1)
        cal = k in STREAM16
        rows.append((k, n, raw, (2 * raw) if cal else raw, w * 1024 / max(nw, 1), int(cal)))
    rows.sort(key=lambda r: -(r[3] + r[4]) * r[1])
    with open(out, "w", newline="") as fh:
        wr = csv.writer(fh)
        wr.writerow(["kernel", "dispatches", "fetch_bytes_per_launch_raw", "fetch_bytes_per_launch_used", "write_bytes_per_launch", "fetch_calibrated_x2"])
        for r in rows:
            if r[0].startswith(("void at::native", "void (anonymous namespace)", "__amd_rocclr")):
                continue            # torch kernels of the synthetic data generator and runtime copies: not the library
            wr.writerow([r[0], r[1], "%.0f" % r[2], "%.0f" % r[3], "%.0f" % r[4], r[5]])
    for r in rows[:25]:
        print("%-40s n=%5d fetch %12.3f MB%s  write %10.3f MB  per launch" % (r[0][:40], r[1], r[3] / 1e6, " (x2)" if r[5] else "     ", r[4] / 1e6))


if __name__ == "__main__":
    main()
