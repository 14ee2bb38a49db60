#!/usr/bin/env python3
"""Matrix-core counters of one rocprofv3 --pmc pass (SQ_INSTS_VALU_MFMA_MOPS_F64, SQ_VALU_MFMA_BUSY_CYCLES, GRBM_GUI_ACTIVE)
per kernel:  python tools/pmc_mfma_summary.py <counter_collection.csv> <out.csv>
f64 MFMA flops = SQ_INSTS_VALU_MFMA_MOPS_F64 x 512 (the counter ticks once per 512 operations); MfmaUtil = busy cycles over
GRBM_GUI_ACTIVE x 1024 SIMDs / 8 XCDs (GRBM_GUI_ACTIVE is summed over the 8 XCDs, MI355X_MICROARCH.md "DVFS")."""
import csv
import sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(float)); cnt = defaultdict(int)
with open(sys.argv[1]) as fh:
    for r in csv.DictReader(fh):
        k = r["Kernel_Name"].split("(")[0]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
            cnt[k] += 1
rows = []
for k, v in acc.items():
    if k.startswith(("void at::native", "__amd_rocclr", "void (anonymous")):
        continue
    n = max(cnt[k], 1)
    gui = v.get("GRBM_GUI_ACTIVE", 0.0) / n; busy = v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / n; mops = v.get("SQ_INSTS_VALU_MFMA_MOPS_F64", 0.0) / n
    util = 100.0 * busy / (gui / 8.0 * 1024.0) if gui > 0 else 0.0
    rows.append((k, n, gui, busy, util, mops * 512.0))
rows.sort(key=lambda r: -r[5] * r[1])
with open(sys.argv[2], "w", newline="") as fh:
    w = csv.writer(fh)
    w.writerow(["kernel", "dispatches", "GRBM_GUI_ACTIVE_per_launch", "SQ_VALU_MFMA_BUSY_CYCLES_per_launch", "MfmaUtil_percent_of_all_1024_SIMDs", "f64_mfma_flops_per_launch(SQ_INSTS_VALU_MFMA_MOPS_F64x512)"])
    for r in rows:
        w.writerow([r[0], r[1], "%.0f" % r[2], "%.0f" % r[3], "%.4f" % r[4], "%.4g" % r[5]])
tot = sum(r[5] * r[1] for r in rows)
for r in rows[:12]:
    print("%-34s n=%5d  f64 MFMA flops/launch %10.4g  share of all MFMA flops %5.1f %%  MfmaUtil %.3f %%" % (r[0][:34], r[1], r[5], 100 * r[5] * r[1] / max(tot, 1), r[4]))
