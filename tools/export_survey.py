#!/usr/bin/env python3
"""Write a synthetic survey as flat binary frame dumps for diasss_amd/host/test_demo.

    frame_%03d.bin = int32 N, int32 M, f64 raw[N*M], f64 pose[N*6], f64 alt[N], f64 gr[M/2]
"""
import os
import sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def export(out_dir, F=3, N=700, M=480, seed=77):
    from diasss_amd.synth import Survey
    os.makedirs(out_dir, exist_ok=True)
    sv = Survey(F, N, M, seed=seed)
    for f in range(F):
        raw = sv.frame(f).cpu().numpy()
        pose, alt, gr = sv.inputs(f)
        with open(os.path.join(out_dir, "frame_%03d.bin" % f), "wb") as fh:
            fh.write(np.array([N, M], np.int32).tobytes())
            for a in (raw, pose, alt, gr):
                fh.write(np.ascontiguousarray(a, np.float64).tobytes())
    return sv


if __name__ == "__main__":
    a = sys.argv
    export(a[1], *(int(v) for v in a[2:6]))
