#!/usr/bin/env python3
"""Timeline of ONE factorisation + solve out of a rocprofv3 kernel trace (the last complete LM trial):
    python tools/trace_trial.py gpurun_out/prof_x/x_kernel_trace.csv [--all]"""
import csv
import sys
from collections import defaultdict
rows = []
with open(sys.argv[1]) as f:
    for x in csv.DictReader(f):
        rows.append((int(x["Start_Timestamp"]), int(x["End_Timestamp"]), x["Kernel_Name"].split("(")[0], int(x["Grid_Size_X"]), int(x["Workgroup_Size_X"]), int(x["Grid_Size_Y"]), int(x["Grid_Size_Z"])))
rows.sort()
idx = [i for i, x in enumerate(rows) if x[2] == "pg_assemble_kernel"]
import os
back = int(os.environ.get("TRIAL_FROM_END", "3"))        # TRIAL_FROM_END=8: a trial of the last TIMED step of a default bench.py run (five trials per step; the last pass is the profiled one)
a, b = idx[-back], idx[-back + 1]
t0 = rows[a][0]
print("one trial span %.3f ms" % ((rows[b][0] - t0) / 1e6))
agg = defaultdict(lambda: [0, 0.0])
prev_end = t0
gaps = 0.0
for x in rows[a:b]:
    agg[x[2]][0] += 1; agg[x[2]][1] += (x[1] - x[0]) / 1e3
    gaps += max(0, x[0] - prev_end) / 1e3; prev_end = max(prev_end, x[1])
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%-32s %5d launches %9.1f us  avg %7.1f" % (k, v[0], v[1], v[1] / v[0]))
print("idle gaps between kernels %.1f us" % gaps)
if "--all" in sys.argv:
    for x in rows[a:b]:
        print("%9.1f %8.1f %-28s grid %d/%d y%d" % ((x[0] - t0) / 1e3, (x[1] - x[0]) / 1e3, x[2], x[3], x[4], x[5]))
