#!/usr/bin/env python3
"""Wall time of frame set-up + extraction (set_frames starts the extraction, extract finishes it; device-resident frames), for A/B runs of its switches
(DSSS_LIB = another build of the library): python tools/extract_only.py [frames=200] [repeats=10]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diasss_amd.synth import Survey
from diasss_amd.pipeline import Pipeline
F = int(sys.argv[1]) if len(sys.argv) > 1 else 200
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
N, M = 2000, 1024
sv = Survey(F, N, M, seed=20240602, device="cuda:0")
raws = [sv.frame(f) for f in range(F)]
ins = [sv.inputs(f) for f in range(F)]
poses = [i[0] for i in ins]; alts = [i[1] for i in ins]; grs = [i[2] for i in ins]
pipe = Pipeline(F)
ts = []
for s in range(reps + 2):
    torch.cuda.synchronize(); pipe.ctx.sync()
    t0 = time.perf_counter()
    pipe.set_frames(raws, poses, alts, grs)                 # (starts the extraction of device-resident frames itself)
    pipe.extract(); pipe.ctx.sync()
    ts.append(1e3 * (time.perf_counter() - t0))
ts = sorted(ts[2:])
print("set_frames + extract, %d frames: median %.3f ms, min %.3f | %s %s" % (F, ts[len(ts) // 2], ts[0], os.environ.get("DSSS_LIB", "<tree>")[-18:],
      " ".join("%s=%s" % (k, v) for k, v in sorted(os.environ.items()) if k.startswith("DSSS_EX_"))))
pipe.close()
