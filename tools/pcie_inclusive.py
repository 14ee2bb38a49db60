"""PCIe-inclusive rate of the hot path: the raw waterfalls start in HOST memory (pageable numpy), so every step uploads
200 x 16.4 MB before it can normalise.  Never the `value` of bench.py (that one starts with the frames resident in
HBM); DESIGN.md section 2 quotes this figure.   python tools/pcie_inclusive.py [C3|C2]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from diasss_amd.pipeline import Pipeline
from diasss_amd.synth import Survey
wl = sys.argv[1] if len(sys.argv) > 1 else "C3"
F, N, M = {"C2": (50, 1000, 512), "C3": (200, 2000, 1024)}[wl]
sv = Survey(F, N, M, seed=20240601 + 1, device="cuda:0")
raws = [sv.frame(f).cpu().numpy() for f in range(F)]          # host resident
ins = [sv.inputs(f) for f in range(F)]
poses, alts, grs = [i[0] for i in ins], [i[1] for i in ins], [i[2] for i in ins]
pipe = Pipeline(F)
for it in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    pipe.run(raws, poses, alts, grs)
    dt = time.perf_counter() - t0
    print("step %d: %.1f ms  -> %.0f frames/s with the upload of %.2f GB of host-resident frames inside the step" % (it, dt * 1e3, F / dt, F * N * M * 8 / 1e9))
