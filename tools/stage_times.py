#!/usr/bin/env python3
"""Wall time of the four pipeline stages of one C3 step, a device synchronisation after each (the bench itself does not
synchronise between stages): where the host-side time of a step sits.  python tools/stage_times.py [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from diasss_amd.synth import Survey
from diasss_amd.pipeline import Pipeline

F, N, M = 200, 2000, 1024
sv = Survey(F, N, M, seed=20240601 + 1, device="cuda:0")
raws = [sv.frame(f) for f in range(F)]
ins = [sv.inputs(f) for f in range(F)]
poses = [i[0] for i in ins]; alts = [i[1] for i in ins]; grs = [i[2] for i in ins]
pipe = Pipeline(F)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 4
acc = np.zeros(5)
for s in range(steps + 1):
    t = [time.perf_counter()]
    pipe.set_frames(raws, poses, alts, grs); torch.cuda.synchronize(); t.append(time.perf_counter())
    pipe.extract(); torch.cuda.synchronize(); t.append(time.perf_counter())
    src_tgt = pipe.match(); torch.cuda.synchronize(); t.append(time.perf_counter())
    pipe.optimize(); torch.cuda.synchronize(); t.append(time.perf_counter())
    if s > 0:
        acc[:4] += np.diff(t); acc[4] += t[-1] - t[0]
print("ms per step: set_frames %.2f  extract %.2f  match+lc %.2f  posegraph %.2f  total %.2f" % tuple(1e3 * acc / steps))
pipe.close()
