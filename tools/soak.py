#!/usr/bin/env python3
"""Soak run of the whole pipeline on one GPU: N steps of BASELINE config 3 on the same survey; the poses must keep their hash and the device
memory in use must not grow (arenas, the matcher's grid, the page-locked mirrors are allocated once).
    python tools/soak.py [steps=400]"""
import hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diasss_amd.synth import Survey
from diasss_amd.pipeline import Pipeline

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 400
F, N, M = 200, 2000, 1024
sv = Survey(F, N, M, seed=20240602, device="cuda:0")
raws = [sv.frame(f) for f in range(F)]
ins = [sv.inputs(f) for f in range(F)]
poses = [i[0] for i in ins]; alts = [i[1] for i in ins]; grs = [i[2] for i in ins]
pipe = Pipeline(F)
h0 = m0 = None
for s in range(steps):
    p, st = pipe.run(raws, poses, alts, grs)
    if s in (5, steps // 4, steps - 1):
        pipe.ctx.sync()
        free, total = torch.cuda.mem_get_info()
        h = hashlib.sha1(p.tobytes()).hexdigest()[:12]
        print("step %d: %.3f GB in use, poses sha %s, LM iterations %d" % (s, (total - free) / 2**30, h, st[0]), flush=True)
        if h0 is None:
            h0, m0 = h, total - free
        else:
            assert h == h0, "the result changed between steps"
            assert abs((total - free) - m0) < 64 * 2**20, "device memory in use grew"
print("soak ok: %d steps" % steps)
pipe.close()
