"""GPU parity of SURVEY.md 8f N4 -- the 128-float descriptor of the reference's SIFT call site
(/root/reference/thirdparty/ORBextractor.cpp:1043-1047,1098) and the L2 branch of the matcher on its rows
(/root/reference/src/core/FEAmatcher.cpp:106-139) -- against the oracle (oracle/orc_sift.c, orc_match.c use_l2 = 2).
The histogram is accumulated in fixed point on both sides, so the bar is BIT-EXACT rows and identical match indices."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture()
def ctx():
    from diasss_amd import capi
    c = capi.Context(max_frames=8)
    yield c
    c.close()


def _sift_mode(ctx, l2=True, **orb):
    from diasss_amd import capi
    mp, op, mt, pg = ctx.default_params()
    op.descriptor = capi.DESC_SIFT128
    for k, v in orb.items():
        setattr(op, k, v)
    mt.use_l2 = 2 if l2 else 0
    ctx.set_params(orb=op, match=mt)
    return op, mt


@pytest.mark.parametrize("shape,kw", [((700, 480), {}), ((640, 400), dict(nfeatures=500, nlevels=4)), ((1000, 512), dict(nfeatures=3000))])
def test_sift_rows_bit_exact_vs_oracle(ctx, orc, shape, kw):
    """every keypoint of every pyramid level, border keypoints included (19 px from the edge: the window is clipped there)"""
    from diasss_amd.synth import Survey
    N, M = shape
    sv = Survey(2, N, M, seed=33)
    _sift_mode(ctx, **kw)
    po = orc.orb_params()
    for k, v in kw.items():
        setattr(po, k, v)
    for f in range(2):
        raw = sv.frame(f).numpy()
        pose, alt, gr = sv.inputs(f)
        ctx.frame_set(f, raw, N, M, pose, alt, gr)
        n = ctx.extract(f)
        kps, desc, geo = ctx.features_get(f)
        d128 = ctx.features_get_sift(f)
        o_kps, o_desc, _, _, o_d128 = orc.detect_feature(raw, None, po, sift=True)
        assert n == len(o_kps) > 100
        assert (desc == o_desc).all() and (kps["angle"] == o_kps["angle"]).all() and (kps["x"] == o_kps["x"]).all()      # the ORB outputs are untouched
        assert d128.dtype == np.float32 and d128.shape == (n, 128)
        bad = np.nonzero((d128 != o_d128.astype(np.float32)).any(1))[0]
        assert len(bad) == 0, "SIFT rows differ at keypoints %s (levels %s)" % (bad[:8], kps["octave"][bad[:8]])
        assert (d128.max(1) > 0).all() and len(np.unique(kps["octave"])) >= 3


def test_sift_histograms_identical_under_the_rounded_bytes(ctx, orc, tmp_path, monkeypatch):
    """The 128 output bytes hide a one-unit difference in the 2^-12 fixed-point histogram almost always (it moves a byte once in a few thousand
    keypoints: that is how the full-size C3 test caught a 1-ulp square root).  DSSS_SIFT_HIST_DUMP hands out the raw int32 histograms of every
    PRE-FILTER keypoint: compared one by one with orc_sift_hist on the oracle's blurred pyramid -- every one of ~13 000 rounded shares per
    keypoint summed to the same integers."""
    import ctypes as C
    from diasss_amd.synth import Survey
    N, M = 1200, 800
    sv = Survey(2, N, M, seed=41)
    _sift_mode(ctx)
    raw = sv.frame(1).numpy(); pose, alt, gr = sv.inputs(1)
    dump = str(tmp_path / "hist.bin")
    monkeypatch.setenv("DSSS_SIFT_HIST_DUMP", dump)
    ctx.frame_set(0, raw, N, M, pose, alt, gr); ctx.extract(0)
    monkeypatch.delenv("DSSS_SIFT_HIST_DUMP")
    H = np.fromfile(dump, np.int32).reshape(-1, 128)
    norm = orc.normalize(raw)
    kps, desc, d128 = orc.orb_extract_sift(norm)                  # pre-filter keypoints, the order of the device's list
    assert 500 < len(kps) <= len(H)
    p = orc.orb_params()
    lr = (C.c_int * 8)(); lc = (C.c_int * 8)()
    orc.lib().orc_orb_level_sizes(N, M, C.byref(p), lr, lc)
    pyr = [norm]
    for l in range(1, p.nlevels):
        nxt = np.zeros((lr[l], lc[l]), np.uint8)
        orc.lib().orc_resize_linear_u8(orc.u8(pyr[-1]), pyr[-1].shape[0], pyr[-1].shape[1], orc.u8(nxt), lr[l], lc[l]); pyr.append(nxt)
    blur = [orc.blur13(im) for im in pyr]
    bad = 0
    for i in range(len(kps)):
        l = int(kps["octave"][i]); s = np.float32(1.0)
        for _ in range(l):
            s = np.float32(s * np.float32(1.2))
        x = kps["x"][i] / s if l else kps["x"][i]; y = kps["y"][i] / s if l else kps["y"][i]
        h = orc.sift_hist(blur[l], int(round(float(x))), int(round(float(y))), float(kps["angle"][i]))
        bad += int((H[i] != h).any())
    assert bad == 0, "%d of %d keypoints have another histogram" % (bad, len(kps))
    assert (H[len(kps):] == 0).all()


def test_sift_extract_many_equals_single_and_mode_switch(ctx, orc):
    """the batched entry point (eager start by dsss_frames_set included) gives the same rows; switching the mode off drops them"""
    from diasss_amd.synth import Survey
    from diasss_amd import capi
    F, N, M = 4, 500, 400
    sv = Survey(F, N, M, seed=35, device="cuda:0")
    raws = [sv.frame(f) for f in range(F)]
    ins = [sv.inputs(f) for f in range(F)]
    _sift_mode(ctx)
    ctx.frames_set(list(range(F)), raws, [N] * F, [M] * F, [i[0] for i in ins], [i[1] for i in ins], [i[2] for i in ins])
    ctx.extract_many(list(range(F)))
    many = [ctx.features_get_sift(f) for f in range(F)]
    for f in range(F):
        o = orc.detect_feature(raws[f].cpu().numpy(), sift=True)
        assert (many[f] == o[4].astype(np.float32)).all()
        ctx.extract(f)
        assert (ctx.features_get_sift(f) == many[f]).all()
    mp, op, mt, pg = ctx.default_params()                       # back to the ORB-only configuration
    ctx.set_params(orb=op, match=mt)
    ctx.extract(0)
    with pytest.raises(capi.DsssError):
        ctx.features_get_sift(0)


def _frames_from_oracle(ctx, orc, F, N, M, seed):
    from diasss_amd.synth import Survey
    sv = Survey(F, N, M, seed=seed)
    fr = {}
    for f in range(F):
        raw = sv.frame(f).numpy()
        pose, alt, gr = sv.inputs(f)
        kps, desc, _, _, d128 = orc.detect_feature(raw, sift=True)
        ctx.frame_set(f, None, N, M, pose, alt, gr)
        ctx.features_set(f, N, M, kps, desc)
        ctx.features_set_sift(f, d128.astype(np.float32))
        fr[f] = dict(N=N, M=M, pose=pose, alt=alt, gr=gr, kps=kps, desc=d128, geo=orc.geo_at_kps(pose, gr, M, kps), bb=orc.geo_bbox(pose, gr, M))
    return fr


@pytest.mark.parametrize("grid", ["1", "0"])
def test_l2_matcher_on_sift_rows_vs_oracle(ctx, orc, grid, monkeypatch):
    """both matcher kernels (geo grid; all pairs with LDS-staged 128-byte tiles, DSSS_MT_GRID=0): first stage, SCC, rows, kp7 identical"""
    from tests.test_gpu_matcher import _check_pair
    monkeypatch.setenv("DSSS_MT_GRID", grid)
    fr = _frames_from_oracle(ctx, orc, 3, 700, 480, 21)
    _sift_mode(ctx)
    po = orc.match_params(); po.use_l2 = 2
    src = [0, 0, 1]; tgt = [1, 2, 2]
    ctx.match_pairs(src, tgt)
    n = [_check_pair(ctx, orc, p, i, j, fr, po) for p, (i, j) in enumerate(zip(src, tgt))]
    assert n[0] > 5 and n[2] > 5, n


def test_l2_matcher_ties_and_sizes(ctx, orc):
    """random integer rows with FEW distinct values (tied minima all the time: lowest index wins, second = best), sizes around the tile
    edges, a frame with no keypoints"""
    from tests import helpers as H
    from tests.test_gpu_matcher import _check_pair
    N, M = 700, 480
    _sift_mode(ctx)
    po = orc.match_params(); po.use_l2 = 2
    for sizes in ((0, 5), (1, 1), (63, 65), (300, 257)):
        fr = {}
        rng = np.random.default_rng(sum(sizes))
        protos = rng.integers(0, 256, (6, 128)).astype(np.uint8)
        for f, n in enumerate(sizes):
            pose, alt, gr = H.track(N, M, 0, seed=3)
            pose = pose.copy(); pose[:, 4] += 0.3 * f
            kps, desc = H.random_features(N, M, n, 50 + f + 10 * n)
            d128 = protos[rng.integers(0, 6, n)].copy()
            d128[rng.random(n) < 0.3, 5] ^= 1                                  # a few rows one step away from their prototype
            ctx.frame_set(2 * f, None, N, M, pose, alt, gr)
            ctx.features_set(2 * f, N, M, kps, desc)
            ctx.features_set_sift(2 * f, d128.astype(np.float32))
            geo = orc.geo_at_kps(pose, gr, M, kps) if n else np.zeros((0, 2))
            fr[2 * f] = dict(N=N, M=M, pose=pose, alt=alt, gr=gr, kps=kps, desc=d128, geo=geo, bb=orc.geo_bbox(pose, gr, M))
        ctx.match_pairs([0], [2])
        _check_pair(ctx, orc, 0, 0, 2, fr, po)


def test_l2_128_needs_the_rows(ctx, orc):
    from diasss_amd import capi
    from tests import helpers as H
    N, M = 700, 480
    for f in range(2):
        pose, alt, gr = H.track(N, M, 0, seed=3)
        kps, desc = H.random_features(N, M, 50, 9 + f)
        ctx.frame_set(f, None, N, M, pose, alt, gr)
        ctx.features_set(f, N, M, kps, desc)
    mp, op, mt, pg = ctx.default_params()
    mt.use_l2 = 2
    ctx.set_params(match=mt)
    with pytest.raises(capi.DsssError):
        ctx.match_pairs([0], [1])
    mt.use_l2 = 3
    with pytest.raises(capi.DsssError):
        ctx.set_params(match=mt)


def test_pipeline_in_sift_mode_vs_oracle(orc):
    """extraction with DSSS_DESC_SIFT128 -> L2 matching on the 128 rows -> reprojection -> mini-LMs -> pose graph, against the oracle run
    the same way: rows and kp7 bit-exact, same loop-closure edges, poses within 1e-6 (a 6-frame survey: the oracle's envelope solver)"""
    from diasss_amd.pipeline import Pipeline, all_pairs
    from diasss_amd.synth import Survey
    from diasss_amd import capi
    F, N, M = 6, 700, 480
    sv = Survey(F, N, M, seed=91)
    raws = [sv.frame(f).numpy() for f in range(F)]
    ins = [sv.inputs(f) for f in range(F)]
    pipe = Pipeline(F, device=0)
    mp, op, mt, pg = pipe.ctx.default_params()
    op.descriptor = capi.DESC_SIFT128; mt.use_l2 = 2
    pipe.ctx.set_params(orb=op, match=mt)
    g_poses, g_stats = pipe.run(raws, [i[0] for i in ins], [i[1] for i in ins], [i[2] for i in ins])
    g_poses = g_poses.copy()
    po = orc.match_params(); po.use_l2 = 2
    fr = []
    for f in range(F):
        pose, alt, gr = ins[f]
        kps, desc, _, _, d128 = orc.detect_feature(raws[f], sift=True)
        fr.append(dict(pose=pose, alt=alt, gr=gr, kps=kps, desc=d128, geo=orc.geo_at_kps(pose, gr, M, kps), bb=orc.geo_bbox(pose, gr, M)))
        assert (pipe.ctx.features_get_sift(f) == d128.astype(np.float32)).all()
    src, tgt = all_pairs(F)
    off = [0]; k7s = []; lcs = []; nrows = 0
    for p in range(len(src)):
        i, j = int(src[p]), int(tgt[p])
        a, b = fr[i], fr[j]
        rows = orc.robust_matching(i, j, N, N, a["kps"], a["desc"], a["geo"], a["bb"], b["kps"], b["desc"], b["geo"], b["bb"], po)
        assert (pipe.ctx.match_rows(p) == rows).all(), "rows of pair %d-%d differ" % (i, j)
        kp7 = orc.get_kps_pairs(rows, j, a["alt"], a["gr"], b["alt"], b["gr"])
        assert (pipe.ctx.match_kp7(p) == kp7).all()
        k7s.append(kp7); lcs.append(orc.lc_solve(kp7, a["pose"], a["alt"], a["gr"], M, b["pose"], b["alt"], b["gr"], M))
        off.append(off[-1] + len(kp7)); nrows += len(rows)
    assert nrows > 30
    edges = orc.pg_select_lc([N] * F, src, tgt, off, np.concatenate(k7s), np.concatenate(lcs))
    g_edges = pipe.ctx.posegraph_select(F)
    assert len(g_edges) == len(edges) > 5 and (g_edges["a"] == edges["a"]).all() and (g_edges["b"] == edges["b"]).all()
    o_poses, o_stats = orc.pg_solve(np.concatenate([i[0] for i in ins]), edges)
    assert o_stats[0] == g_stats[0] and np.abs(g_poses - o_poses).max() < 1e-6
    pipe.close()


def test_sift_mode_on_frames_without_keypoints(ctx, orc):
    """a flat frame (no FAST corner anywhere) and a frame whose keypoints all fall under the mask: zero rows, no fault; matching such frames in
    L2-128 mode gives empty rows"""
    from tests import helpers as H
    N, M = 700, 480
    _sift_mode(ctx)
    pose, alt, gr = H.track(N, M, 0, seed=3)
    flat = np.full((N, M), 7.0)
    ctx.frame_set(0, flat, N, M, pose, alt, gr)
    assert ctx.extract(0) == 0
    assert ctx.features_get_sift(0).shape == (0, 128)
    rng = np.random.default_rng(1)
    speck = rng.rayleigh(1.0, (N, M))
    pose2 = pose.copy(); pose2[:, 4] += 0.3
    ctx.frame_set(2, speck, N, M, pose2, alt, gr)
    n2 = ctx.extract(2)
    o = orc.detect_feature(speck, sift=True)
    assert n2 == len(o[0]) and (ctx.features_get_sift(2) == o[4].astype(np.float32)).all()
    ctx.match_pairs([0], [2])
    assert len(ctx.match_rows(0)) == 0

