#!/usr/bin/env python3
"""Instruction-issue counters of one rocprofv3 --pmc pass (SQ_INSTS_VALU, SQ_INSTS_SALU, SQ_INSTS_LDS, SQ_WAVES) per kernel,
joined with the kernel durations of the same run's kernel trace:
    python tools/pmc_issue_summary.py <counter_collection.csv> <kernel_trace.csv> <out.csv>
VALU issue utilisation = SQ_INSTS_VALU / duration against the MEASURED issue rate of the full-rate instruction class at 8
wavefronts per SIMD, chip-wide (tools/ubench/valu_issue.hip, profiles/r03_ubench_valu_issue.txt: v_fma_f32 1015 G wave-instructions/s;
the same constant bench.py uses).  The half-rate classes -- packed 16-bit min / max, v_bcnt, v_dot4, v_perm, three-operand integer
operations, every f64 operation -- issue at 490-570 G/s, so a kernel made of them (FAST, descriptors, matcher, mini-LM) saturates at
about 0.5 on this scale; round 2 priced everything at one instruction per 4 cycles (614 G/s), i.e. 1.65 x these figures."""
import csv
import sys
from collections import defaultdict

VALU_ISSUE_PER_S = 1015e9
acc = defaultdict(lambda: defaultdict(float)); cnt = defaultdict(int)
with open(sys.argv[1]) as fh:
    for r in csv.DictReader(fh):
        k = r["Kernel_Name"].split("(")[0]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_WAVES":
            cnt[k] += 1
dur = defaultdict(float); nd = defaultdict(int)
with open(sys.argv[2]) as fh:
    for r in csv.DictReader(fh):
        k = r["Kernel_Name"].split("(")[0]
        dur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9; nd[k] += 1
rows = []
for k, v in acc.items():
    if k.startswith(("void at::native", "__amd_rocclr", "void (anonymous")):
        continue
    n = max(cnt[k], 1)
    t = dur[k] / max(nd[k], 1)
    valu = v.get("SQ_INSTS_VALU", 0.0) / n; salu = v.get("SQ_INSTS_SALU", 0.0) / n; lds = v.get("SQ_INSTS_LDS", 0.0) / n; waves = v.get("SQ_WAVES", 0.0) / n
    util = valu / t / VALU_ISSUE_PER_S if t > 0 else 0.0
    rows.append((k, n, t * 1e6, waves, valu, salu, lds, util))
rows.sort(key=lambda r: -r[2] * r[1])
with open(sys.argv[3], "w", newline="") as fh:
    w = csv.writer(fh)
    w.writerow(["kernel", "dispatches", "avg_us_under_counters", "waves_per_launch", "SQ_INSTS_VALU_per_launch", "SQ_INSTS_SALU_per_launch", "SQ_INSTS_LDS_per_launch", "valu_issue_utilisation"])
    for r in rows:
        w.writerow([r[0], r[1], "%.2f" % r[2], "%.0f" % r[3], "%.4g" % r[4], "%.4g" % r[5], "%.4g" % r[6], "%.4f" % r[7]])
for r in rows[:16]:
    print("%-30s n=%5d %9.1f us  waves %9.0f  VALU/wave %8.0f  LDS/wave %7.0f  SALU/wave %7.0f  VALU issue utilisation %5.1f %%" %
          (r[0][:30], r[1], r[2], r[3], r[4] / max(r[3], 1), r[6] / max(r[3], 1), r[5] / max(r[3], 1), 100 * r[7]))
