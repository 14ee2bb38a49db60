#!/usr/bin/env python3
"""Every kernel / copy of the LAST step of a rocprofv3 kernel trace of bench.py, in start order: offset, duration, idle gap before it.
    python tools/trace_step.py <kernel_trace.csv> [first_kernel_prefix=row_reduce_kernel] [from_us] [to_us]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
first = sys.argv[2] if len(sys.argv) > 2 else "row_reduce_kernel"
lo = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
hi = float(sys.argv[4]) if len(sys.argv) > 4 else 1e18
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith(first)]
i0 = starts[-1]
t0 = int(rows[i0]["Start_Timestamp"])
end = t0
for r in rows[i0:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    off = (s - t0) / 1e3
    if lo <= off <= hi:
        gap = (s - end) / 1e3
        print("%10.1f %8.1f %s%s" % (off, (e - s) / 1e3, ("[idle %7.1f] " % gap) if gap > 5 else "               ", r["Kernel_Name"].split("(")[0][:60]))
    end = max(end, e)
