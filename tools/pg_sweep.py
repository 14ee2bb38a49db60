#!/usr/bin/env python3
"""Time the pose-graph solve alone on a dumped loop-closure edge set (tools/dump_graph.py), for A/B runs of the
analysis knobs (DSSS_PG_BIN_COST, DSSS_PG_ND_BOTH, DSSS_PG_LEAF: they are read per solve, but the worker pool is per process, so
one process per setting).
    python tools/pg_sweep.py gpurun_out/C3_edges.npz [repeats]
Prints: median / minimum wall time of dsss_posegraph_solve_edges, LM iterations and final error."""
import os
import sys
import time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from diasss_amd import capi                      # noqa: E402
from diasss_amd.synth import Survey              # noqa: E402

d = np.load(sys.argv[1])
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 7
N, F = int(d["N"]), int(d["F"])
sv = Survey(F, N, 1024 if N == 2000 else 512, seed=20240601 + (1 if N == 2000 else 0))
dr = np.concatenate([sv.inputs(f)[0] for f in range(F)])
edges = np.zeros(len(d["a"]), capi.LCEDGE_DTYPE)
for k in ("a", "b", "rel", "var"):
    edges[k] = d[k]
c = capi.Context(max_frames=2)
ts = []
for r in range(reps + 1):
    t0 = time.perf_counter(); p, s = c.posegraph_solve_edges(dr, edges); ts.append((time.perf_counter() - t0) * 1e3)
ts = sorted(ts[1:])
if not os.environ.get("PG_SWEEP_NOPROF"):             # (the event scopes of the profiling mode put ~10 us between launches: not under a kernel trace)
    c.profile(True); c.profile_reset()
    c.posegraph_solve_edges(dr, edges)
    prof = {k: (round(v[0], 3), v[1]) for k, v in c.profile_get().items() if v[1] > 0}
    c.profile(False)
    print("  kernel families of one solve (ms, launches):", prof)
import hashlib
print("solve ms median %.2f min %.2f | iterations %d error %.6f | poses sha %s | %s" % (ts[len(ts) // 2], ts[0], s[0], s[2], hashlib.sha1(p.tobytes()).hexdigest()[:12],
      " ".join("%s=%s" % (k, v) for k, v in sorted(os.environ.items()) if k.startswith("DSSS_PG_") and k != "DSSS_PG_VERBOSE")))
c.close()
