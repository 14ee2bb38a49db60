#!/usr/bin/env python3
"""Two surveys in flight on one GPU: two contexts (own streams), two host threads, every thread runs whole steps of the hot
path on the same HBM-resident input.  The pose-graph solve of one survey is latency-bound and leaves the chip idle; the
extraction of the next one fills it.  Prints ms per step for 1 and 2 jobs in flight (throughput, not latency)."""
import os
import sys
import threading
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch                                      # noqa: E402
from bench import WORKLOADS                       # noqa: E402
from diasss_amd.pipeline import Pipeline          # noqa: E402
from diasss_amd.synth import Survey               # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "C3"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
wl = WORKLOADS[name]
F, N, M = wl["F"], wl["N"], wl["M"]
sv = Survey(F, N, M, seed=20240601 + sorted(WORKLOADS).index(name), device="cuda:0")
raws = [sv.frame(f) for f in range(F)]
ins = [sv.inputs(f) for f in range(F)]
poses = [i[0] for i in ins]; alts = [i[1] for i in ins]; grs = [i[2] for i in ins]
torch.cuda.synchronize()
for jobs in (1, 2, 3):
    pipes = [Pipeline(F) for _ in range(jobs)]
    for p in pipes:
        p.run(raws, poses, alts, grs)
    res = [None] * jobs

    def work(k):
        for _ in range(steps):
            res[k] = p_run(pipes[k])

    def p_run(p):
        return p.run(raws, poses, alts, grs)[1]
    th = [threading.Thread(target=work, args=(k,)) for k in range(jobs)]
    t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    for p in pipes: p.ctx.sync()
    dt = time.perf_counter() - t0
    print("%d job(s) in flight: %.2f ms per step, %.0f frames/s, stats %s" % (jobs, 1e3 * dt / (steps * jobs), F * steps * jobs / dt, [float(s) for s in res[0]]))
    for p in pipes: p.close()
