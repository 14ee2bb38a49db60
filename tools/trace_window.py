#!/usr/bin/env python3
"""Kernels of the LAST step of a rocprofv3 kernel trace of bench.py between two kernel-name patterns: python tools/trace_window.py <kernel_trace.csv> <from> <to>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("row_reduce_kernel")]
i0 = starts[-int(__import__("os").environ.get("STEP_FROM_END", "1"))]          # STEP_FROM_END=2: the last TIMED step of a default bench.py run (the very last pass is the profiled one)
t0 = int(rows[i0]["Start_Timestamp"])
on = False; prev_end = None
_k = int(__import__("os").environ.get("STEP_FROM_END", "1"))
_i1 = starts[-_k + 1] if _k > 1 else len(rows)
for r in rows[i0:_i1]:
    nm = r["Kernel_Name"]
    if not on and sys.argv[2] in nm:
        on = True
    if on:
        s = (int(r["Start_Timestamp"]) - t0) / 1e3; e = (int(r["End_Timestamp"]) - t0) / 1e3
        gap = "" if prev_end is None else "  (+%.1f idle)" % (s - prev_end) if s - prev_end > 5 else ""
        print("%9.1f %8.1f  %s%s" % (s, e - s, nm[:70], gap))
        prev_end = max(prev_end or e, e)
        if sys.argv[3] in nm:
            break
