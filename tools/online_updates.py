#!/usr/bin/env python3
"""N3 at the size of BASELINE config 3: the survey fed frame by frame, global updates (dsss_posegraph_update: the whole graph re-analysed and
re-factorised, warm-started) against incremental ones (dsss_posegraph_update_window: the last W frames, conditioned on the frozen rest)
followed by ONE global update.  Prints the cost per update along the survey and where the estimates end.
    python tools/online_updates.py [frames=200] [window=3] [pings=2000] [bins=1024]"""
import os
import sys
import time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from diasss_amd.pipeline import Pipeline         # noqa: E402
from diasss_amd.synth import Survey              # noqa: E402

F = int(sys.argv[1]) if len(sys.argv) > 1 else 200
W = int(sys.argv[2]) if len(sys.argv) > 2 else 3
N = int(sys.argv[3]) if len(sys.argv) > 3 else 2000
M = int(sys.argv[4]) if len(sys.argv) > 4 else 1024
sv = Survey(F, N, M, seed=20240601 + 1, device="cuda:0")
raws = [sv.frame(f) for f in range(F)]
ins = [sv.inputs(f) for f in range(F)]
pipe = Pipeline(F, device=0)
b_out, b_stats = pipe.run(raws, [i[0] for i in ins], [i[1] for i in ins], [i[2] for i in ins])
b_out = b_out.copy()
ctx = pipe.ctx
n_edges = len(ctx.posegraph_select(F))
src, tgt = pipe.src, pipe.tgt
act = [p for p in range(len(src)) if ctx.pair_is_active(p)]
kp7 = {p: ctx.match_kp7(p) for p in act}
by_tgt = {}
for p in act:
    if len(kp7[p]):
        by_tgt.setdefault(int(tgt[p]), []).append(p)
ctx.posegraph_solve(F, F * N, want_rpy=False)
t0 = time.perf_counter(); ctx.posegraph_solve(F, F * N, want_rpy=False); t_batch = time.perf_counter() - t0


def online(window):
    ctx.posegraph_reset()
    ts, tl = [], []
    for j in range(F):
        pj = by_tgt.get(j, [])
        t1 = time.perf_counter()
        if pj:
            ctx.lc_solve_pairs([src[p] for p in pj], [tgt[p] for p in pj], [kp7[p] for p in pj])
        ctx.sync(); t2 = time.perf_counter()
        if window:
            ctx.posegraph_update_window(j + 1, (j + 1) * N, window, want_poses=False)
        else:
            ctx.L.dsss_posegraph_update(ctx.h, j + 1, None, None, None)
        ts.append(time.perf_counter() - t2); tl.append(t2 - t1)
    return 1e3 * np.array(ts), 1e3 * np.array(tl)


online(W)
tw, tlw = online(W)
w_out, _ = ctx.posegraph_update_window(F, F * N, W)
t1 = time.perf_counter(); p_out, _, p_stats = ctx.posegraph_update(F, F * N); t_polish = time.perf_counter() - t1
tg, _ = online(0)
g_out, _, g_stats = ctx.posegraph_update(F, F * N)
q = [slice(0, F // 4), slice(F // 4, F // 2), slice(F // 2, 3 * F // 4), slice(3 * F // 4, F)]
print("%d frames of %d x %d, %d loop closures; one batch solve %.1f ms (%d LM iterations)" % (F, N, M, n_edges, 1e3 * t_batch, b_stats[0]))
print("global updates      : %.1f ms in all, per update by quarter of the survey %s ms" % (tg.sum(), " ".join("%.2f" % tg[s].mean() for s in q)))
print("window updates (W=%d): %.1f ms in all, per update by quarter of the survey %s ms (+ %.2f ms of mini-LMs per frame); final global update %.1f ms, %d LM iterations"
      % (W, tw.sum(), " ".join("%.2f" % tw[s].mean() for s in q), tlw.mean(), 1e3 * t_polish, p_stats[0]))
print("max |position - batch|: windowed estimate %.3g m, after the final global update %.3g m, global updates %.3g m; objectives: batch %.6e, windowed+final %.6e, global %.6e"
      % (np.abs(w_out[:, 9:] - b_out[:, 9:]).max(), np.abs(p_out[:, 9:] - b_out[:, 9:]).max(), np.abs(g_out[:, 9:] - b_out[:, 9:]).max(), b_stats[2], p_stats[2], g_stats[2]))
pipe.close()
