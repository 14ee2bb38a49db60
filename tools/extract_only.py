import os, sys
sys.path.insert(0, "/root/repo")
import torch
from diasss_amd.synth import Survey
from diasss_amd.pipeline import Pipeline
F, N, M = 50, 2000, 1024
sv = Survey(F, N, M, seed=20240602, device="cuda:0")
raws = [sv.frame(f) for f in range(F)]
ins = [sv.inputs(f) for f in range(F)]
pipe = Pipeline(F)
for s in range(2):
    pipe.set_frames(raws, [i[0] for i in ins], [i[1] for i in ins], [i[2] for i in ins]); pipe.extract(); torch.cuda.synchronize()
pipe.close()
