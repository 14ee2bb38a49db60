#!/usr/bin/env python3
"""Timeline of the LAST extraction in a rocprofv3 kernel trace of tools/extract_only.py: python tools/trace_extract.py <kernel_trace.csv>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last row_reduce launch starts the last extraction
starts = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("row_reduce_kernel")]
i0 = starts[-int(__import__("os").environ.get("STEP_FROM_END", "1"))]          # STEP_FROM_END=2: the last TIMED step of a default bench.py run (the very last pass is the profiled one)
t0 = int(rows[i0]["Start_Timestamp"])
_k = int(__import__("os").environ.get("STEP_FROM_END", "1"))
_i1 = starts[-_k + 1] if _k > 1 else len(rows)
for r in rows[i0:_i1]:
    s = (int(r["Start_Timestamp"]) - t0) / 1e3; e = (int(r["End_Timestamp"]) - t0) / 1e3
    print("%9.1f %9.1f %8.1f  q%-3s %s" % (s, e, e - s, r.get("Queue_Id", "?"), r["Kernel_Name"][:60]))
