#!/usr/bin/env python3
"""Idle gaps of the GPU inside the LAST step of a rocprofv3 kernel trace of bench.py: python tools/trace_gaps.py <kernel_trace.csv> [min_gap_us=20]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
mingap = float(sys.argv[2]) if len(sys.argv) > 2 else 20.0
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("row_reduce_kernel")]
i0 = starts[-int(__import__("os").environ.get("STEP_FROM_END", "1"))]          # STEP_FROM_END=2: the last TIMED step of a default bench.py run (the very last pass is the profiled one)
t0 = int(rows[i0]["Start_Timestamp"])
busy_end = t0; busy = 0.0; gaps = []
last = rows[i0]["Kernel_Name"]
_k = int(__import__("os").environ.get("STEP_FROM_END", "1"))
_i1 = starts[-_k + 1] if _k > 1 else len(rows)
for r in rows[i0:_i1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if s > busy_end:
        gaps.append(((s - busy_end) / 1e3, (busy_end - t0) / 1e3, last[:40], r["Kernel_Name"][:40]))
        busy += 0
    if e > busy_end:
        busy += (e - max(s, busy_end)) / 1e3
        busy_end = e; last = r["Kernel_Name"]
span = (busy_end - t0) / 1e3
print("last step: span %.1f us, GPU busy %.1f us, idle %.1f us in %d gaps" % (span, busy, span - busy, len(gaps)))
small = sum(g[0] for g in gaps if g[0] < mingap)
print("gaps below %.0f us: %.1f us in total" % (mingap, small))
for g in sorted(gaps, reverse=True)[:25]:
    if g[0] >= mingap:
        print("%9.1f us idle at %9.1f   after %-40s before %s" % g)
