"""GPU parity: batched mini-LM loop-closure solve, LC selection and the pose-graph LM against the CPU oracle.
Floating point: tolerances are written next to each assertion (north_star: 1e-6 on optimised poses)."""
import os
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from diasss_amd import capi
    c = capi.Context(max_frames=8)
    yield c
    c.close()


@pytest.fixture(scope="module")
def survey(ctx, orc):
    """3 legs, features from the oracle extractor imported into the context, all pairs matched"""
    from diasss_amd.synth import Survey
    F, N, M = 3, 700, 480
    sv = Survey(F, N, M, seed=31)
    fr = []
    for f in range(F):
        raw = sv.frame(f).numpy()
        pose, alt, gr = sv.inputs(f)
        kps, desc, _, _ = orc.detect_feature(raw)
        ctx.frame_set(f, None, N, M, pose, alt, gr)
        ctx.features_set(f, N, M, kps, desc)
        fr.append(dict(N=N, M=M, pose=pose, alt=alt, gr=gr, kps=kps, desc=desc, geo=orc.geo_at_kps(pose, gr, M, kps), bb=orc.geo_bbox(pose, gr, M)))
    src = [0, 0, 1]; tgt = [1, 2, 2]
    ctx.match_pairs(src, tgt)
    ctx.lc_solve_all()
    return dict(F=F, N=N, M=M, fr=fr, src=src, tgt=tgt)


def _oracle_lc(orc, sv):
    out = []
    for i, j in zip(sv["src"], sv["tgt"]):
        a, b = sv["fr"][i], sv["fr"][j]
        rows = orc.robust_matching(i, j, a["N"], b["N"], a["kps"], a["desc"], a["geo"], a["bb"], b["kps"], b["desc"], b["geo"], b["bb"])
        kp7 = orc.get_kps_pairs(rows, j, a["alt"], a["gr"], b["alt"], b["gr"])
        lcs = orc.lc_solve(kp7, a["pose"], a["alt"], a["gr"], a["M"], b["pose"], b["alt"], b["gr"], b["M"])
        out.append((kp7, lcs))
    return out


def test_lc_batch_parity(ctx, orc, survey):
    ref = _oracle_lc(orc, survey)
    total = 0
    for p, (kp7, lcs) in enumerate(ref):
        g = ctx.lc_get(p)
        assert len(g) == len(lcs)
        if not len(g):
            continue
        total += len(g)
        assert (g["iters"] == lcs["iters"]).all()                       # same LM path
        assert np.allclose(g["rel"], lcs["rel"], rtol=0, atol=1e-9)     # relative pose: 1e-9 (rad / m)
        assert np.allclose(g["var"], lcs["var"], rtol=1e-6, atol=1e-15)
        assert np.allclose(g["score"], lcs["score"], rtol=0, atol=1e-6)
        assert np.allclose(g["err1"], lcs["err1"], rtol=1e-6, atol=1e-12)
    assert total > 50


def test_lc_standalone_and_sticky_flip(ctx, orc, survey):
    """dsss_lc_solve with caller kp7: pair (0,1) has opposite headings so the yaw compensation (optimizer.cpp:697-703) is active"""
    a, b = survey["fr"][0], survey["fr"][1]
    kp7 = ctx.match_kp7(0)
    assert len(kp7) > 5
    g = ctx.lc_solve(0, 1, kp7)
    o = orc.lc_solve(kp7, a["pose"], a["alt"], a["gr"], a["M"], b["pose"], b["alt"], b["gr"], b["M"])
    assert np.allclose(g["rel"], o["rel"], rtol=0, atol=1e-9) and (g["iters"] == o["iters"]).all()


def test_lc_selection_and_posegraph_parity(ctx, orc, survey):
    F, N = survey["F"], survey["N"]
    ref = _oracle_lc(orc, survey)
    pair_off = [0]
    for kp7, _ in ref:
        pair_off.append(pair_off[-1] + len(kp7))
    kp7_all = np.concatenate([r[0] for r in ref]); lcs_all = np.concatenate([r[1] for r in ref])
    o_edges = orc.pg_select_lc([N] * F, survey["src"], survey["tgt"], pair_off, kp7_all, lcs_all)
    g_edges = ctx.posegraph_select(F)
    assert len(g_edges) == len(o_edges) and len(o_edges) > 10
    assert (g_edges["a"] == o_edges["a"]).all() and (g_edges["b"] == o_edges["b"]).all()        # selection rule is exact
    assert np.allclose(g_edges["rel"], o_edges["rel"], atol=1e-9)
    dr = np.concatenate([f["pose"] for f in survey["fr"]])
    o_out, o_stats = orc.pg_solve(dr, o_edges)
    g_out, g_rpy, g_stats = ctx.posegraph_solve(F, F * N)
    assert g_stats[0] == o_stats[0]                                                             # same number of LM iterations
    assert np.allclose(g_stats[2], o_stats[2], rtol=1e-6)
    assert np.abs(g_out - o_out).max() < 1e-6                                                   # north_star: poses within 1e-6
    assert np.abs(g_rpy[:, 3:] - o_out[:, 9:]).max() < 1e-6


def test_posegraph_edges_api_small_cases(ctx, orc):
    """explicit chains: no LC; one LC; LC between adjacent poses (coincides with a chain coupling); long interior segments; the SAME two
    poses closed twice (two factors on one block: both count, as in the oracle's dense normal equations -- a plain += in the scatter lost one
    of them now and then) and once more in the other direction"""
    n = 300
    rng = np.random.default_rng(3)
    dr = np.zeros((n, 6)); dr[:, 3] = 0.05 * np.arange(n); dr[:, 2] = 0.01 * np.sin(np.arange(n) / 30.0)
    dr[150:, 2] += 3.14159265359; dr[150:, 4] += 5.0; dr[150:, 3] = dr[149, 3] - 0.05 * np.arange(150)
    def edge(a, b, dy):
        e = np.zeros(1, orc.LCEDGE_DTYPE)
        e["a"] = a; e["b"] = b
        Ta = orc.Pose(); Tb = orc.Pose(); Tr = orc.Pose()
        import ctypes as C
        orc.lib().orc_pose_from_rodrigues(orc.dp(np.ascontiguousarray(dr[a])), C.byref(Ta))
        orc.lib().orc_pose_from_rodrigues(orc.dp(np.ascontiguousarray(dr[b])), C.byref(Tb))
        orc.lib().orc_pose_between(C.byref(Ta), C.byref(Tb), C.byref(Tr))
        rel = np.concatenate([np.array(Tr.R), np.array(Tr.t)]); rel[10] += dy
        e["rel"][0] = rel; e["var"][0] = [1e-6, 1e-6, 1e-5, 1e-3, 0.5, 1e-2]
        return e
    cases = [np.zeros(0, orc.LCEDGE_DTYPE), edge(20, 280, 0.3), np.concatenate([edge(10, 290, 0.2), edge(60, 240, -0.1), edge(149, 150, 0.05), edge(100, 151, 0.1)]),
             np.concatenate([edge(10, 290, 0.2), edge(10, 290, 0.25), edge(60, 240, -0.1), edge(240, 60, 0.08)]),
             # three and four closures on ONE pair, in both directions and out of order, and two on a pair of NEIGHBOURING separators
             # (their block already holds the chain coupling): the scatter sums a group in edge order with one writer per element
             np.concatenate([edge(240, 60, 0.08), edge(10, 290, 0.2), edge(60, 240, -0.1), edge(290, 10, 0.22), edge(10, 290, 0.25),
                             edge(60, 240, -0.05), edge(149, 150, 0.05), edge(150, 149, -0.02), edge(10, 290, 0.21)])]
    for edges in cases:
        o_out, o_stats = orc.pg_solve(dr, edges)
        g_out, g_stats = ctx.posegraph_solve_edges(dr, edges)
        assert g_stats[0] == o_stats[0]
        assert np.abs(g_out - o_out).max() < 1e-6
        for _ in range(3):                                   # same bits every run (duplicates used to be added atomically)
            g2, s2 = ctx.posegraph_solve_edges(dr, edges)
            assert (g2 == g_out).all() and (np.asarray(s2) == np.asarray(g_stats)).all()


def test_posegraph_error_exits_leave_the_context_usable(orc):
    """the error exits of the solve (an edge out of range before anything runs; a variance that is not positive or a relative pose that is not
    finite AFTER the analysis thread has started and the preparation is in flight) return an error, and the next solve on the
    same context gives the bits of a fresh one: the exit joins the analysis thread and drains its stream before the arena goes back"""
    from diasss_amd import capi
    from tests.test_gpu_configs import _lawnmower_graph
    dr, gt, edges = _lawnmower_graph(6, 3000, 800, seed=5)
    c = capi.Context(max_frames=2)
    ref, sref = c.posegraph_solve_edges(dr, edges)
    for env in ({}, {"DSSS_PG_ND_INDEX": "0"}):                            # with and without the chain-order cut candidate
        old = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        try:
            for kind in ("range", "var0", "varnan", "relnan", "var0"):
                bad = edges.copy()
                k = len(bad) // 2
                if kind == "range": bad["b"][k] = len(dr) + 5
                elif kind == "var0": bad["var"][k, 3] = 0.0
                elif kind == "varnan": bad["var"][k, 1] = np.nan
                else: bad["rel"][k, 4] = np.inf
                with pytest.raises(capi.DsssError):
                    c.posegraph_solve_edges(dr, bad)
                p, st = c.posegraph_solve_edges(dr, edges)
                if not env:
                    assert (p == ref).all() and (np.asarray(st) == np.asarray(sref)).all(), kind
                else:
                    assert st[0] == sref[0] and np.abs(p - ref).max() < 1e-8, kind
        finally:
            for k, v in old.items():
                if v is None: os.environ.pop(k, None)
                else: os.environ[k] = v
    c.close()


def test_initial_values_follow_libstdcxx_normal_stream(ctx, orc):
    """max_iters = 0 returns the initial estimate DR o noise (optimizer.cpp:150-160): the device reproduces the
    sequential minstd_rand0 + polar normal_distribution stream with a jump-ahead per attempt"""
    n = 5000
    rng = np.random.default_rng(4)
    dr = np.zeros((n, 6)); dr[:, 3] = 0.05 * np.arange(n); dr[:, 2] = rng.uniform(-3, 3, n); dr[:, 5] = rng.uniform(-1, 1, n)
    mp, op, mt, pg = ctx.default_params()
    pg.max_iters = 0
    ctx.set_params(pg=pg)
    po = orc.pg_params(); po.max_iters = 0
    g_out, _ = ctx.posegraph_solve_edges(dr, np.zeros(0, orc.LCEDGE_DTYPE))
    o_out, _ = orc.pg_solve(dr, np.zeros(0, orc.LCEDGE_DTYPE), po)
    pg.max_iters = 100
    ctx.set_params(pg=pg)
    z = np.zeros(6 * n); orc.lib().orc_normal_fill(orc.dp(z), 6 * n)
    assert np.abs(o_out[:, 9:] - dr[:, 3:]).max() > 0.5            # noise really applied (sigma 0.5 m)
    assert np.abs(g_out - o_out).max() < 1e-12


def test_online_updates_reach_the_batch_optimum(ctx, orc, survey):
    """N3 (optimizer.cpp:134-272, the iSAM2 loop): frame-by-frame dsss_posegraph_update -- each update consumes the loop
    closures that END in the new frame and starts from the previous estimate -- against ONE batch solve of the same graph.
    Both are LM iterates stopped by the same relative tolerance (1e-5 on the objective) near the same minimum, so the
    comparison is on the objective (oracle-evaluated at both results), not bit for bit on the poses."""
    F, N = survey["F"], survey["N"]
    src, tgt = survey["src"], survey["tgt"]
    b_edges = ctx.posegraph_select(F)
    b_out, _, b_stats = ctx.posegraph_solve(F, F * N, want_rpy=False)
    kp7 = [ctx.match_kp7(p) for p in range(len(src))]
    try:
        ctx.posegraph_reset()                      # marks the LC set of the fixture as consumed: the graph starts empty
        steps = []
        for j in range(F):
            pj = [p for p in range(len(src)) if tgt[p] == j and len(kp7[p])]
            if pj:
                ctx.lc_solve_pairs([src[p] for p in pj], [tgt[p] for p in pj], [kp7[p] for p in pj])
            o_out, _, o_stats = ctx.posegraph_update(j + 1, (j + 1) * N)
            steps.append(int(o_stats[0]))
        assert ctx.posegraph_online_edges() == len(b_edges)
        dr = np.concatenate([f["pose"] for f in survey["fr"]])
        e_batch = orc.pg_error_at(dr, b_edges, b_out); e_online = orc.pg_error_at(dr, b_edges, o_out)
        assert np.isclose(e_batch, b_stats[2], rtol=1e-6) and np.isclose(e_online, o_stats[2], rtol=1e-6)
        assert e_online <= e_batch * (1 + 1e-3), (e_online, e_batch)
        assert np.abs(o_out[:, 9:] - b_out[:, 9:]).max() < 5e-3                  # metres: same minimum to the LM stopping tolerance
        # an update that adds nothing converges at once: one more call, no new LC set, at most one accepted step
        again, _, st = ctx.posegraph_update(F, F * N)
        assert st[0] <= 1 and np.abs(again - o_out).max() < 1e-3
        # ... and the online run ENDS AT THE BATCH OPTIMUM, not merely near it: with the stopping tolerances tightened for both
        # (the LM then runs until the objective stops changing in the 12th digit) the last update and a batch solve of the same
        # graph agree to a tenth of a micrometre and to 1e-9 on the objective
        mp_, op_, mt_, pg_ = ctx.default_params()
        pg_.rel_tol = 1e-13; pg_.abs_tol = 1e-13; pg_.max_iters = 100
        ctx.set_params(pg=pg_)
        try:
            tight_o, _, st_o = ctx.posegraph_update(F, F * N)
            tight_b, st_b = ctx.posegraph_solve_edges(dr, b_edges)           # the batch solve of the accumulated graph (explicit edges: the context holds only the last LC set)
        finally:
            mp_, op_, mt_, pg_ = ctx.default_params()
            ctx.set_params(pg=pg_)
        e_b = orc.pg_error_at(dr, b_edges, tight_b); e_o = orc.pg_error_at(dr, b_edges, tight_o)
        assert abs(e_o - e_b) <= 1e-9 * e_b, (e_o, e_b)
        assert np.abs(tight_o[:, 9:] - tight_b[:, 9:]).max() < 1e-7 and np.abs(tight_o[:, :9] - tight_b[:, :9]).max() < 1e-9
        # shrinking the graph under accumulated edges is refused, not silently wrong
        with pytest.raises(Exception):
            ctx.posegraph_update(1, N)
        # the same through the matcher itself, frame by frame (the call sequence of INTEGRATION.md): frame j is matched against
        # the earlier frames, its loop closures are solved, the graph is updated
        ctx.posegraph_reset()
        for j in range(F):
            pj = [p for p in range(len(src)) if tgt[p] == j]
            if pj:
                ctx.match_pairs([src[p] for p in pj], [tgt[p] for p in pj]); ctx.lc_solve_all()
            m_out, _, m_stats = ctx.posegraph_update(j + 1, (j + 1) * N)
        assert ctx.posegraph_online_edges() == len(b_edges)
        assert np.abs(m_out - o_out).max() < 1e-9 and m_stats[0] == o_stats[0]      # same edges, same warm starts: the same run
    finally:
        ctx.posegraph_reset()
        ctx.match_pairs(src, tgt); ctx.lc_solve_all()


def test_incremental_window_updates(orc):
    """N3, the incremental form (dsss_posegraph_update_window; optimizer.cpp:134-139,262-272 keeps ONE ISAM2 object across pings): a survey
    fed frame by frame, every update solving only the last 3 frames conditioned on the frozen estimate of the rest.
      - the cost of an update does not grow with the survey (the global update's does): timed and printed, asserted loosely;
      - the windowed estimate is a good one on its own (objective within a few per cent of the batch optimum);
      - ONE global update at the end, warm-started from it, lands on the batch optimum (objective 1e-9 relative, positions 1e-7 m with the
        stopping tolerances tightened), in fewer LM iterations than a cold batch solve -- what the reference reads after its loop (:279);
      - the whole online run costs less than three batch solves."""
    import time
    from diasss_amd import capi
    from diasss_amd.pipeline import Pipeline, all_pairs
    from diasss_amd.synth import Survey
    F, N, M, W = 24, 700, 480, 3
    sv = Survey(F, N, M, seed=77, device="cuda:0")
    raws = [sv.frame(f) for f in range(F)]
    ins = [sv.inputs(f) for f in range(F)]
    pipe = Pipeline(F, device=0)
    b_out, b_stats = pipe.run(raws, [i[0] for i in ins], [i[1] for i in ins], [i[2] for i in ins])      # the batch run: features, matches, mini-LMs, solve
    b_out = b_out.copy(); b_stats = np.array(b_stats)
    ctx = pipe.ctx
    b_edges = ctx.posegraph_select(F)
    assert len(b_edges) > 60
    src, tgt = pipe.src, pipe.tgt
    kp7 = [ctx.match_kp7(p) for p in range(len(src))]
    dr = np.concatenate([i[0] for i in ins])
    t0 = time.perf_counter(); ctx.posegraph_solve(F, F * N, want_rpy=False); t_batch = time.perf_counter() - t0

    def online(window):
        ctx.posegraph_reset()
        ts, its = [], []
        for j in range(F):
            pj = [p for p in range(len(src)) if tgt[p] == j and len(kp7[p])]
            if pj:
                ctx.lc_solve_pairs([src[p] for p in pj], [tgt[p] for p in pj], [kp7[p] for p in pj])
            ctx.sync(); t1 = time.perf_counter()
            if window:
                _, st = ctx.posegraph_update_window(j + 1, (j + 1) * N, window, want_poses=False)
            else:
                _, _, st = ctx.posegraph_update(j + 1, (j + 1) * N)
            ts.append(time.perf_counter() - t1); its.append(int(st[0]))
        return np.array(ts), its
    online(W)                                                  # (first use: allocations)
    tw, iw = online(W)
    assert ctx.posegraph_online_edges() == len(b_edges)
    w_out, _ = ctx.posegraph_update_window(F, F * N, W)        # nothing new: the window once more, and the whole trajectory comes back
    e_b = orc.pg_error_at(dr, b_edges, b_out); e_w = orc.pg_error_at(dr, b_edges, w_out)
    t1 = time.perf_counter(); p_out, _, p_stats = ctx.posegraph_update(F, F * N); t_polish = time.perf_counter() - t1
    e_p = orc.pg_error_at(dr, b_edges, p_out)
    tg, ig = online(0)
    late_w, late_g = tw[F // 2:].mean(), tg[F // 2:].mean()
    print("incremental updates, %d frames of %d pings, window %d: per update %.2f ms (first half %.2f, second half %.2f); global updates %.2f ms (%.2f -> %.2f); "
          "one batch solve %.2f ms; online run %.1f ms windowed + %.2f ms final global update vs %.1f ms of global updates; objective batch %.6e, windowed %.6e, "
          "after the final update %.6e (%d LM iterations; cold batch: %d)"
          % (F, N, W, 1e3 * tw.mean(), 1e3 * tw[:F // 2].mean(), 1e3 * late_w, 1e3 * tg.mean(), 1e3 * tg[:F // 2].mean(), 1e3 * late_g, 1e3 * t_batch,
             1e3 * tw.sum(), 1e3 * t_polish, 1e3 * tg.sum(), e_b, e_w, e_p, p_stats[0], b_stats[0]))
    assert late_w < 0.8 * late_g                              # a window does not pay for the frames behind it
    assert tw[F // 2:].mean() < 1.5 * tw[W + 1:F // 2].mean() # ... and does not grow with the survey
    assert e_w < 1.10 * e_b and np.abs(w_out[:, 9:] - b_out[:, 9:]).max() < 0.5
    assert e_p <= e_b * (1 + 1e-3) and p_stats[0] <= b_stats[0]
    # tightened: the final global update ends AT the batch optimum
    mp_, op_, mt_, pg_ = ctx.default_params()
    pg_.rel_tol = 1e-13; pg_.abs_tol = 1e-13; pg_.max_iters = 100
    ctx.set_params(pg=pg_)
    tight_o, _, _ = ctx.posegraph_update(F, F * N)
    tight_b, _ = ctx.posegraph_solve_edges(dr, b_edges)
    e_o = orc.pg_error_at(dr, b_edges, tight_o); e_t = orc.pg_error_at(dr, b_edges, tight_b)
    assert abs(e_o - e_t) <= 1e-9 * e_t and np.abs(tight_o[:, 9:] - tight_b[:, 9:]).max() < 1e-7
    pipe.close()


def test_window_update_edge_cases(ctx, orc, survey):
    """dsss_posegraph_update_window at its borders: a window that covers the whole survey IS the global update (same bits); a window of one
    frame; an update that brings no new loop closures; window_frames < 1 is refused"""
    from diasss_amd import capi
    F, N = survey["F"], survey["N"]
    src, tgt = survey["src"], survey["tgt"]
    kp7 = [ctx.match_kp7(p) for p in range(len(src))]

    def feed(j):
        pj = [p for p in range(len(src)) if tgt[p] == j and len(kp7[p])]
        if pj:
            ctx.lc_solve_pairs([src[p] for p in pj], [tgt[p] for p in pj], [kp7[p] for p in pj])
    try:
        ctx.posegraph_reset()
        for j in range(F):
            feed(j)
            g_out, _, g_st = ctx.posegraph_update(j + 1, (j + 1) * N)
        ctx.posegraph_reset()
        for j in range(F):
            feed(j)
            w_out, w_st = ctx.posegraph_update_window(j + 1, (j + 1) * N, F + 3)      # never a frozen part: the global path
        assert (w_out == g_out).all() and (w_st == g_st).all()
        ctx.posegraph_reset()
        for j in range(F):
            feed(j)
            o1, s1 = ctx.posegraph_update_window(j + 1, (j + 1) * N, 1)               # one frame at a time, everything before it frozen
        assert o1.shape == (F * N, 12) and np.isfinite(o1).all()
        dr = np.concatenate([f["pose"] for f in survey["fr"]])
        edges = ctx.posegraph_select(F)
        # (frozen frames never move again: further from the optimum than a wider window, but a sane trajectory -- within 2 m of the batch one on this survey)
        assert np.abs(o1[:, 9:] - g_out[:, 9:]).max() < 2.0
        again, s2 = ctx.posegraph_update_window(F, F * N, 2)                           # nothing new: at most one accepted step, nothing moves far
        assert s2[0] <= 2 and np.abs(again - o1)[: (F - 2) * N].max() == 0             # ... and the frozen part is bit for bit what it was
        with pytest.raises(capi.DsssError):
            ctx.posegraph_update_window(F, F * N, 0)
    finally:
        ctx.posegraph_reset()

