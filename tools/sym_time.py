#!/usr/bin/env python3
"""Host-only timing of the pose-graph analysis (pg_symbolic + the launch lists), no GPU: the dumped C3 graph or a synthetic lawn-mower
graph of C5's size, analysed as the product does it (bins' lists left to the device), several repetitions in one process.
    python tools/sym_time.py c3|c5 [repetitions] [threads]
DSSS_PG_VERBOSE=1 prints the phases of the last repetition."""
import ctypes as C
import os
import sys
import time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
which = sys.argv[1] if len(sys.argv) > 1 else "c3"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
if len(sys.argv) > 3:
    os.environ["DSSS_SYM_THREADS"] = sys.argv[3]
os.environ.setdefault("DSSS_PG_BIN_COST", "600")
from diasss_amd import capi                      # noqa: E402
nparts = int(os.environ.get("SYM_PARTS", "1"))       # partitions of contiguous frame blocks (dsss_set_pg_partitions)
part = None
if which in ("c3", "c5real"):
    from diasss_amd.synth import Survey
    d = np.load(os.path.join(ROOT, "tools", "_data", "C3_edges.npz" if which == "c3" else "C5_edges.npz"))
    a, b, N, F = d["a"], d["b"], int(d["N"]), int(d["F"])
    n = N * F
    sv = Survey(F, N, 1024 if which == "c3" else 2048, seed=20240601 + (1 if which == "c3" else 3))
    dr = np.concatenate([sv.inputs(f)[0] for f in range(F)])
    is_sep = np.zeros(n, bool); is_sep[0] = is_sep[-1] = True; is_sep[a] = True; is_sep[b] = True
    pb = [(F * q // nparts) * N for q in range(nparts)] + [n]
    if nparts > 1 and os.environ.get("SYM_MOVE", "1") != "0":          # the boundaries move to their cheapest cuts (pg_solve_impl)
        ends = np.nonzero(is_sep)[0]
        lo = np.searchsorted(ends, np.minimum(a, b)); hi = np.searchsorted(ends, np.maximum(a, b))
        cross = np.zeros(len(ends) + 2, np.int64); np.add.at(cross, lo + 1, 1); np.add.at(cross, hi + 1, -1); cross = np.cumsum(cross)
        width = n // nparts // 3
        for q in range(1, nparts):
            target = pb[q]; best = None
            for i in range(max(1, np.searchsorted(ends, target - width)), len(ends)):
                start = ends[i - 1] + 1
                if start > target + width: break
                if start <= pb[q - 1] or start < target - width: continue
                key = (cross[i], abs(start - target))
                if best is None or key < best[0]: best = (key, start)
            if best is not None: pb[q] = int(best[1])
        print("partition boundaries", pb[1:-1], "crossing loop closures", [int(cross[np.searchsorted(ends, x - 1) + 1]) for x in pb[1:-1]])
    for q in range(1, nparts):
        is_sep[pb[q] - 1] = True
    # gaps of 16 chunks or more get separators of their own (pg_solve_impl): the first chunk end at least 256 poses after the last one
    marked = np.nonzero(is_sep)[0]
    fill = []
    for lo, hi in zip(marked[:-1], marked[1:]):
        last = lo
        while True:
            nx = (last + 256 + 15) // 16 * 16
            if nx >= hi: break
            fill.append(nx); last = nx
    is_sep[np.array(fill, np.int64)] = True
    sep = np.nonzero(is_sep)[0]
    if nparts > 1:
        part = np.ascontiguousarray((np.searchsorted(np.array(pb), sep, side="right") - 1).astype(np.int32))
    sidx = -np.ones(n, np.int64); sidx[sep] = np.arange(len(sep))
    ns = len(sep)
    ea = np.concatenate([np.arange(ns - 1), sidx[a]]).astype(np.int32)
    eb = np.concatenate([np.arange(1, ns), sidx[b]]).astype(np.int32)
    cx = np.ascontiguousarray(dr[sep, 3]); cy = np.ascontiguousarray(dr[sep, 4])
else:                                            # 1000 legs x 635 separators, chords between neighbouring legs: the size of C5's reduced graph
    rng = np.random.default_rng(5)
    legs, per = 1000, 635
    ns = legs * per
    k = np.arange(ns); l = k // per; i = k % per
    cx = np.where(l % 2 == 0, i, per - 1 - i).astype(np.float64); cy = 3.0 * l + 0.1 * rng.standard_normal(ns)
    ca, cb = [], []
    for dl, dens in ((1, 0.45), (2, 0.1)):
        src = k[l + dl < legs]
        src = src[rng.random(len(src)) < dens]
        ls, xs = src // per, cx[src]
        tgt = (ls + dl) * per + np.where((ls + dl) % 2 == 0, xs, per - 1 - xs).astype(np.int64)
        ca.append(src); cb.append(tgt)
    ea = np.concatenate([np.arange(ns - 1)] + ca).astype(np.int32); eb = np.concatenate([np.arange(1, ns)] + cb).astype(np.int32)
st = np.zeros(8, np.int64)
p = lambda x: x.ctypes.data_as(C.c_void_p)
ts = []
for r in range(reps):
    if r == reps - 1 and os.environ.get("DSSS_PG_VERBOSE"):
        pass
    t0 = time.perf_counter()
    rc = capi.lib().dsss_host_pg_solve(ns, p(ea), p(eb), len(ea), p(cx), p(cy), p(part) if part is not None else None, nparts, None, None, None, p(st))
    ts.append((time.perf_counter() - t0) * 1e3)
    assert rc == 0
print("%s: ns %d edges %d parts %d | analysis + launch lists ms: %s | median %.2f | nnzL fronts panels levels front_doubles comm_doubles binned maxfront: %s" % (which, ns, len(ea), nparts, " ".join("%.1f" % t for t in ts), sorted(ts)[len(ts) // 2], st.tolist()))
