#!/usr/bin/env python3
"""Distances between LM trajectories of ONE pose graph (a dumped loop-closure edge set, tools/dump_graph.py) under different linear solvers:
the device (1 and 8 partitions), the oracle's LM with its reduced system through scipy's sparse LU (oracle/binding.py: solver="sparse"),
with and without iterative refinement -- at GTSAM's default stopping rule and run to convergence (rel_tol = abs_tol = 1e-13).
    python tools/pg_parity.py tools/_data/C3_edges.npz
Test infrastructure (imports the oracle): how the tolerances of tests/test_gpu_configs.py / test_gpu_multirank.py were chosen."""
import os
import sys
import time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from diasss_amd import capi                      # noqa: E402
from diasss_amd.synth import Survey              # noqa: E402
from oracle import binding as orc                # noqa: E402

d = np.load(sys.argv[1])
N, F = int(d["N"]), int(d["F"])
sv = Survey(F, N, 1024 if N == 2000 else 512, seed=20240601 + (1 if N == 2000 else 0))
dr = np.concatenate([sv.inputs(f)[0] for f in range(F)])
edges = np.zeros(len(d["a"]), capi.LCEDGE_DTYPE)
o_edges = np.zeros(len(d["a"]), orc.LCEDGE_DTYPE)
for k in ("a", "b", "rel", "var"):
    edges[k] = d[k]; o_edges[k] = d[k]
for tol in (None, 1e-13):
    runs = {}
    for parts in (1, 8):
        c = capi.Context(max_frames=2)
        if parts > 1:
            c.set_pg_partitions(parts)
        if tol is not None:
            pg = c.default_params()[3]; pg.rel_tol = tol; pg.abs_tol = tol; c.set_params(pg=pg)
        p, s = c.posegraph_solve_edges(dr, edges)
        runs["device/%d" % parts] = (p.copy(), np.array(s)); c.close()
    po = orc.pg_params()
    if tol is not None:
        po.rel_tol = tol; po.abs_tol = tol
    for name, kw in (("oracle/lu", dict(refine=0)), ("oracle/lu+refine2", dict(refine=2))):
        t0 = time.time()
        p, s = orc.pg_solve(dr, o_edges, po, solver="sparse", **kw)
        runs[name] = (p, s); print("  %s: %.0f s" % (name, time.time() - t0), flush=True)
    names = list(runs)
    print("stopping rule:", "GTSAM default" if tol is None else "rel_tol = abs_tol = %g" % tol)
    for n in names:
        print("  %-18s iterations %d error %.9e -> %.12e" % (n, runs[n][1][0], runs[n][1][1], runs[n][1][2]))
    print("  max |pose difference| (12 numbers per pose: R row-major, t):")
    for i, a in enumerate(names):
        print("    %-18s" % a + " ".join("%9.2e" % np.abs(runs[a][0] - runs[b][0]).max() for b in names[:i + 1]), flush=True)
