#!/usr/bin/env python3
"""Timeline of the LAST extraction in a rocprofv3 kernel trace of tools/extract_only.py: python tools/trace_extract.py <kernel_trace.csv>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last row_reduce launch starts the last extraction
starts = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("row_reduce_kernel")]
i0 = starts[-1]
t0 = int(rows[i0]["Start_Timestamp"])
for r in rows[i0:]:
    s = (int(r["Start_Timestamp"]) - t0) / 1e3; e = (int(r["End_Timestamp"]) - t0) / 1e3
    print("%9.1f %9.1f %8.1f  q%-3s %s" % (s, e, e - s, r.get("Queue_Id", "?"), r["Kernel_Name"][:60]))
