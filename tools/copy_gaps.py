import csv, sys, glob
f = glob.glob(sys.argv[1] + "/*memory_copy_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
print(rows[0].keys())
h2d = [r for r in rows if "HOST_TO_DEVICE" in r.get("Direction", r.get("Kind", "")) ]
print(len(rows), len(h2d))
big = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in h2d if int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) > 150000]
big.sort()
print("big copies", len(big))
# last 400 big copies = 2 steps
b = big[-400:]
durs = [e - s for s, e in b]; gaps = [b[i + 1][0] - b[i][1] for i in range(len(b) - 1)]
import statistics as st
print("dur us median %.1f min %.1f max %.1f" % (st.median(durs) / 1e3, min(durs) / 1e3, max(durs) / 1e3))
g = sorted(gaps)
print("gaps us: median %.1f p90 %.1f max %.1f, sum of gaps within steps %.1f ms" % (st.median(g) / 1e3, g[int(.9 * len(g))] / 1e3, g[-1] / 1e3, sum(x for x in g if x < 5e6) / 1e6))
print("span of last 200: %.1f ms" % ((b[-1][1] - b[-200][0]) / 1e6))
