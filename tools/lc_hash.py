#!/usr/bin/env python3
"""Hash of every loop-closure measurement of a workload (bit-reproducibility / A-B of two builds: DSSS_LIB=<other.so>).
    python tools/lc_hash.py [C2|C3]"""
import hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bench import WORKLOADS
from diasss_amd.synth import Survey
from diasss_amd.pipeline import Pipeline

name = sys.argv[1] if len(sys.argv) > 1 else "C2"
wl = WORKLOADS[name]
F, N, M = wl["F"], wl["N"], wl["M"]
sv = Survey(F, N, M, seed=20240601 + ["C2", "C3"].index(name), device="cuda:0")
raws = [sv.frame(f) for f in range(F)]
ins = [sv.inputs(f) for f in range(F)]
pipe = Pipeline(F)
pipe.set_frames(raws, [i[0] for i in ins], [i[1] for i in ins], [i[2] for i in ins])
pipe.extract(); pipe.match()
h = hashlib.sha1(); n = 0; iters = 0
for p in range(len(pipe.src)):
    if not pipe.ctx.pair_is_active(p):
        continue
    lc = pipe.ctx.lc_get(p)
    h.update(lc.tobytes()); n += len(lc); iters += int(lc["iters"].sum())
print("%s: %d loop-closure problems, %d LM iterations, sha1 %s | %s" % (name, n, iters, h.hexdigest()[:16], os.environ.get("DSSS_LIB", "<tree>")[-20:]))
pipe.close()
