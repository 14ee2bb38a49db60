#!/usr/bin/env python3
"""Where the PCIe-inclusive step goes (upload inside the metric): the bare copy rate of the survey's page-locked frames (one stream,
back to back), the step with host-resident frames by stage, and the same under other upload batch sizes.
    python tools/pcie_probe.py [batch sizes ...]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from diasss_amd.pipeline import Pipeline
from diasss_amd.synth import Survey

F, N, M = 200, 2000, 1024
sv = Survey(F, N, M, seed=20240601 + 1, device="cuda:0")
dev = [sv.frame(f) for f in range(F)]
host = [d.cpu().pin_memory() for d in dev]
ins = [sv.inputs(f) for f in range(F)]
poses, alts, grs = [i[0] for i in ins], [i[1] for i in ins], [i[2] for i in ins]
GB = F * N * M * 8 / 1e9
st = torch.cuda.Stream()
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    with torch.cuda.stream(st):
        for f in range(F):
            dev[f].copy_(host[f], non_blocking=True)
    st.synchronize()
    dt = time.perf_counter() - t0
print("bare upload of %d page-locked frames (%.2f GB), one stream: %.1f ms = %.1f GB/s" % (F, GB, 1e3 * dt, GB / dt))
big = torch.empty((F, N, M), dtype=torch.float64).pin_memory()
dbig = torch.empty((F, N, M), dtype=torch.float64, device="cuda")
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter(); dbig.copy_(big, non_blocking=True); torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("one %.2f GB copy: %.1f ms = %.1f GB/s" % (GB, 1e3 * dt, GB / dt))
del big, dbig
pipe = Pipeline(F)
res = pipe.prepare(dev, poses, alts, grs)
for _ in range(2):
    pipe.run(res)
torch.cuda.synchronize(); t0 = time.perf_counter(); pipe.run(res); torch.cuda.synchronize()
print("resident step: %.1f ms" % (1e3 * (time.perf_counter() - t0)))
hst = pipe.prepare(host, poses, alts, grs)
for b in [None] + [int(a) for a in sys.argv[1:]]:
    if b is not None:
        os.environ["DSSS_EX_UPLOAD_BATCH"] = str(b)
    pipe.run(hst)
    ts = []
    for _ in range(3):
        torch.cuda.synchronize(); t = [time.perf_counter()]
        pipe.set_frames(hst); t.append(time.perf_counter())
        pipe.extract(); pipe.ctx.sync(); t.append(time.perf_counter())
        pipe.match(); pipe.ctx.sync(); t.append(time.perf_counter())
        pipe.optimize(); t.append(time.perf_counter())
        ts.append(np.diff(t) * 1e3)
    m = np.median(np.array(ts), 0)
    print("host-resident frames, upload batch %s: set_frames %.1f | extract (upload inside) %.1f = %.1f GB/s | match + LC %.1f | pose graph %.1f | step %.1f ms"
          % (b if b is not None else "default (8)", m[0], m[1], GB / (m[1] * 1e-3), m[2], m[3], m.sum()))
os.environ.pop("DSSS_EX_UPLOAD_BATCH", None)
for name, arg in (("prepared", hst), ("lists", None)):
    ts = []
    for _ in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        if arg is not None: pipe.run(arg)
        else: pipe.run(host, poses, alts, grs)
        torch.cuda.synchronize(); ts.append(1e3 * (time.perf_counter() - t0))
    print("whole step, host-resident frames (%s): %s ms" % (name, " ".join("%.1f" % t for t in ts)))
h2 = [d.cpu().pin_memory() for d in dev]           # pinned buffers allocated late, as bench.py does
ts = []
for _ in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter(); pipe.run(h2, poses, alts, grs); torch.cuda.synchronize(); ts.append(1e3 * (time.perf_counter() - t0))
print("whole step, frames pinned late: %s ms" % " ".join("%.1f" % t for t in ts))
torch.cuda.synchronize(); pipe.ctx.sync(); t0 = time.perf_counter()
for _ in range(4):
    pipe.run(h2, poses, alts, grs)
torch.cuda.synchronize(); pipe.ctx.sync()
print("four steps back to back, no synchronisation in between: %.1f ms per step" % (1e3 * (time.perf_counter() - t0) / 4))
pipe.close()
