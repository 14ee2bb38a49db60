#!/usr/bin/env python3
"""Offline analysis of a dumped loop-closure edge set (tools/dump_graph.py): separators, ordering, fronts, schedule.
    python tools/pg_stats.py gpurun_out/C3_edges.npz [chunk] [nparts]"""
import ctypes as C
import os
import sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from diasss_amd import capi                      # noqa: E402
from diasss_amd.synth import Survey              # noqa: E402

d = np.load(sys.argv[1])
chunk = int(sys.argv[2]) if len(sys.argv) > 2 else 16
nparts = int(sys.argv[3]) if len(sys.argv) > 3 else 1
a, b, N, F = d["a"], d["b"], int(d["N"]), int(d["F"])
n = N * F
sv = Survey(F, N, 1024 if N == 2000 else 512, seed=20240601 + (1 if N == 2000 else 0))
dr = np.concatenate([sv.inputs(f)[0] for f in range(F)])
# the TRUE separators of the two-level chain elimination (dsss_pg.hip, pg_solve_impl): LC-touched poses, both ends, partition
# ends, and one pose every 16 chunks inside gaps of 16 chunks or more
is_sep = np.zeros(n, bool); is_sep[0] = is_sep[-1] = True; is_sep[a] = True; is_sep[b] = True
fpr = (F + nparts - 1) // nparts
if nparts > 1:
    for r in range(1, nparts):
        is_sep[r * fpr * N - 1] = True
if chunk > 0:
    last = 0
    for i in range(n):
        if is_sep[i]:
            last = i
        elif i - last >= 16 * chunk and i % chunk == 0:
            is_sep[i] = True; last = i
sep = np.nonzero(is_sep)[0]
sidx = -np.ones(n, np.int64); sidx[sep] = np.arange(len(sep))
ns = len(sep)
ea = np.concatenate([np.arange(ns - 1), sidx[a]]).astype(np.int32)
eb = np.concatenate([np.arange(1, ns), sidx[b]]).astype(np.int32)
cx = np.ascontiguousarray(dr[sep, 3]); cy = np.ascontiguousarray(dr[sep, 4])
part = np.ascontiguousarray((sep // N // fpr).astype(np.int32))
st = np.zeros(8, np.int64)
p = lambda x: x.ctypes.data_as(C.c_void_p)
os.environ.setdefault("DSSS_PG_VERBOSE", "1")
rc = capi.lib().dsss_host_pg_solve(ns, p(ea), p(eb), len(ea), p(cx), p(cy), p(part) if nparts > 1 else None, nparts, None, None, None, p(st))
print("rc", rc, "ns", ns, "stats [nnzL fronts panels levels front_doubles comm_doubles binned maxfront]", st.tolist())
