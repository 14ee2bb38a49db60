"""Wall-clock time of each stage of the hot path (frames, extract, match+LC, pose graph) with a device sync between
stages; for use through gpurun:  python tools/stage_times.py C3"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from diasss_amd.pipeline import Pipeline
from diasss_amd.synth import Survey
wl = sys.argv[1] if len(sys.argv) > 1 else "C2"
F, N, M = {"C2": (50, 1000, 512), "C3": (200, 2000, 1024), "C3s": (40, 2000, 1024)}[wl]
sv = Survey(F, N, M, seed=20240601, device="cuda:0")
raws = [sv.frame(f) for f in range(F)]
poses = [sv.inputs(f)[0] for f in range(F)]; alts = [sv.inputs(f)[1] for f in range(F)]; grs = [sv.inputs(f)[2] for f in range(F)]
pipe = Pipeline(F)
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    pipe.set_frames(raws, poses, alts, grs); pipe.ctx.sync(); t1 = time.perf_counter()
    pipe.extract(); pipe.ctx.sync(); t2 = time.perf_counter()
    pipe.match(); pipe.ctx.sync(); t3 = time.perf_counter()
    out, stats = pipe.optimize(); t4 = time.perf_counter()
    print("frames %.1f ms  extract %.1f ms  match+lc %.1f ms  pg %.1f ms  total %.1f ms" % ((t1-t0)*1e3, (t2-t1)*1e3, (t3-t2)*1e3, (t4-t3)*1e3, (t4-t0)*1e3), "rows", pipe.ctx.match_total(), stats)
