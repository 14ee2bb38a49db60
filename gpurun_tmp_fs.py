import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from diasss_amd.pipeline import Pipeline
from diasss_amd.synth import Survey
F, N, M = 200, 2000, 1024
sv = Survey(F, N, M, seed=1, device="cuda:0")
raws = [torch.zeros((N, M), dtype=torch.float64, device="cuda") for f in range(F)]
poses = [sv.inputs(f)[0] for f in range(F)]; alts = [sv.inputs(f)[1] for f in range(F)]; grs = [sv.inputs(f)[2] for f in range(F)]
pipe = Pipeline(F)
for it in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    pipe.set_frames(raws, poses, alts, grs); t1 = time.perf_counter()
    pipe.ctx.sync(); t2 = time.perf_counter()
    print("set_frames call %.2f ms, sync %.2f ms" % ((t1-t0)*1e3, (t2-t1)*1e3))
