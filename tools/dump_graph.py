#!/usr/bin/env python3
"""Run one workload on the GPU and save its loop-closure edge set (gpurun_out/<workload>_edges.npz): the input of the
pose-graph solver, so that ordering / symbolic work can be studied offline on the CPU (the DR poses come from the seeded
Survey generator)."""
import os
import sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import WORKLOADS                                     # noqa: E402
from diasss_amd.pipeline import Pipeline                        # noqa: E402
from diasss_amd.synth import Survey                             # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "C3"
wl = WORKLOADS[name]
F, N, M = wl["F"], wl["N"], wl["M"]
big = F * N * M > (1 << 32)
sv = Survey(F, N, M, seed=20240601 + ["C2", "C3", "smoke", "C5"].index(name), device="cuda:0", noise_on_device=big)      # bench.py's seeds and generator
raws = [sv.frame(f) for f in range(F)]
ins = [sv.inputs(f) for f in range(F)]
pipe = Pipeline(F, nfeatures=wl.get("nfeatures"))
poses, stats = pipe.run(raws, [i[0] for i in ins], [i[1] for i in ins], [i[2] for i in ins])
edges = pipe.ctx.posegraph_select(F, cap=1 << 22)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
extra = {} if big else dict(rel=edges["rel"], var=edges["var"])      # (C5: 351 k edges -- the end points are what the analysis needs; the measurements would be 50 MB)
np.savez_compressed(os.path.join(ROOT, "gpurun_out", "%s_edges.npz" % name), a=edges["a"], b=edges["b"], stats=np.array(stats), N=N, F=F, **extra)
print(name, "edges", len(edges), "stats", stats)
pipe.close()
