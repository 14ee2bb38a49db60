"""GPU parity: the HIP matcher (libdsss.so through the C ABI) against the CPU oracle, bit-exact.
Covers FEAmatcher::GeoNearNeighSearch both stages, ConsistentCheck, RobustMatching rows and GetKpsPairs."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from diasss_amd import capi
    c = capi.Context(max_frames=8)
    yield c
    c.close()


def _oracle_pair(orc, i, j, fr, params=None):
    a, b = fr[i], fr[j]
    d01 = orc.match_dir(i, j, b["N"], a["kps"], a["desc"], a["geo"], b["kps"], b["desc"], b["geo"], b["bb"], params)
    d10 = orc.match_dir(j, i, a["N"], b["kps"], b["desc"], b["geo"], a["kps"], a["desc"], a["geo"], a["bb"], params)
    rows = orc.robust_matching(i, j, a["N"], b["N"], a["kps"], a["desc"], a["geo"], a["bb"], b["kps"], b["desc"], b["geo"], b["bb"], params)
    kp7 = orc.get_kps_pairs(rows, j, a["alt"], a["gr"], b["alt"], b["gr"])
    return d01, d10, rows, kp7


def _check_pair(ctx, orc, p, i, j, fr, params=None):
    d01, d10, rows, kp7 = _oracle_pair(orc, i, j, fr, params)
    for d, ref, f in ((0, d01, i), (1, d10, j)):
        n = len(fr[f]["kps"])
        nn, co, hist, cnt, model = ctx.match_dir(p, d)
        assert (nn[:n] == ref["nn"]).all(), "first-stage CorresID differs (pair %d dir %d)" % (p, d)
        assert (co[:n] == ref["corres"]).all(), "SCC CorresID differs (pair %d dir %d)" % (p, d)
        assert hist == ref["hist"] and cnt == ref["scc_count"] and model == ref["scc_model"]
    g_rows = ctx.match_rows(p)
    assert g_rows.shape == rows.shape and (g_rows == rows).all()
    g_kp7 = ctx.match_kp7(p)
    assert g_kp7.shape == kp7.shape and (g_kp7 == kp7).all()
    return len(rows)


def _mkframes(orc, ctx, sizes, N=700, M=480, seed=0, same_landmarks=True):
    from tests import helpers as H
    fr = []
    base = None
    for f, n in enumerate(sizes):
        pose, alt, gr = H.track(N, M, f, seed=seed)
        if same_landmarks and base is not None and n == len(base[0]):
            kps, desc = base[0].copy(), base[1].copy()
            rng = np.random.default_rng(seed + f)
            for i in range(n):
                for b in rng.choice(256, rng.integers(0, 14), replace=False):
                    desc[i, b // 8] ^= np.uint8(1 << (b % 8))
            # neighbouring leg: mirror rows (opposite heading) and shift bins so the geo points nearly coincide
            if f % 2 == 1:
                kps["y"] = (N - 1) - kps["y"] + rng.integers(-2, 3, n)
                kps["x"] = np.clip((M - 1) - kps["x"] + int(round(0.39 * M)) * 0 + rng.integers(-2, 3, n), 50, M - 50)
            kps["y"] = np.clip(kps["y"], 50, N - 50)
        else:
            kps, desc = H.random_features(N, M, n, seed * 100 + f)
            if n > 0 and base is None:
                base = (kps.copy(), desc.copy())
        ctx.frame_set(f, None, N, M, pose, alt, gr)
        ctx.features_set(f, N, M, kps, desc)
        k2, d2, geo = ctx.features_get(f)
        ogeo = orc.geo_at_kps(pose, gr, M, kps) if n else np.zeros((0, 2))
        assert (geo == ogeo).all(), "device geo lookup differs from the oracle"
        bb = ctx.frame_bbox(f)
        obb = orc.geo_bbox(pose, gr, M)
        assert (bb == obb).all(), "device geo bbox differs from the oracle's full minMaxLoc scan"
        fr.append(dict(N=N, M=M, pose=pose, alt=alt, gr=gr, kps=kps, desc=desc, geo=ogeo, bb=obb))
    return fr


@pytest.mark.parametrize("sizes", [(0, 5), (1, 1), (63, 64), (65, 300), (300, 300), (2000, 2000)])
def test_matcher_sizes(ctx, orc, sizes):
    """legs 0 and 0-copy: same heading, zero spacing, so every keypoint has gate candidates"""
    from tests import helpers as H
    N, M = 700, 480
    fr = []
    for f, n in enumerate(sizes):
        pose, alt, gr = H.track(N, M, 0, seed=3)      # both frames on leg 0: fully overlapping geo boxes
        pose = pose.copy(); pose[:, 4] += 0.3 * f
        kps, desc = H.random_features(N, M, n, 50 + f + 10 * n)
        if f == 1 and n == sizes[0] and n > 0:
            kps = fr[0]["kps"].copy(); desc = fr[0]["desc"].copy()
            rng = np.random.default_rng(n)
            for i in range(n):
                for b in rng.choice(256, rng.integers(0, 20), replace=False):
                    desc[i, b // 8] ^= np.uint8(1 << (b % 8))
            kps["y"] += rng.integers(-3, 4, n).astype(np.float32)
        ctx.frame_set(2 * f, None, N, M, pose, alt, gr)      # ids 0 and 2: same parity (bound 88, no flip)
        ctx.features_set(2 * f, N, M, kps, desc)
        geo = orc.geo_at_kps(pose, gr, M, kps) if n else np.zeros((0, 2))
        fr.append(dict(N=N, M=M, pose=pose, alt=alt, gr=gr, kps=kps, desc=desc, geo=geo, bb=orc.geo_bbox(pose, gr, M)))
    frames = {0: fr[0], 2: fr[1]}
    ctx.match_pairs([0], [2])
    nrows = _check_pair(ctx, orc, 0, 0, 2, frames)
    if sizes == (300, 300) or sizes == (2000, 2000):
        assert nrows > 20


def test_matcher_opposite_heading_and_l2(ctx, orc):
    """odd/even ids: Hamming bound 80, flipped along-track model, img_diff merge term; then the L2-on-bytes mode"""
    from tests import helpers as H
    N, M, n = 700, 480, 400
    pose0, alt0, gr0 = H.track(N, M, 0, seed=5)
    pose1, alt1, gr1 = H.track(N, M, 1, seed=5, spacing=0.0)     # same swath, opposite heading
    k0, d0 = H.random_features(N, M, n, 77)
    k1, d1 = k0.copy(), d0.copy()
    rng = np.random.default_rng(9)
    k1["y"] = np.clip((N - 1) - k0["y"] + rng.integers(-1, 2, n), 60, N - 60).astype(np.float32)
    k1["x"] = np.clip((M - 1) - k0["x"] + rng.integers(-1, 2, n), 60, M - 60).astype(np.float32)
    for i in range(n):
        for b in rng.choice(256, rng.integers(0, 16), replace=False):
            d1[i, b // 8] ^= np.uint8(1 << (b % 8))
    fr = {}
    for f, (pose, alt, gr, k, d) in enumerate(((pose0, alt0, gr0, k0, d0), (pose1, alt1, gr1, k1, d1))):
        ctx.frame_set(f, None, N, M, pose, alt, gr)
        ctx.features_set(f, N, M, k, d)
        fr[f] = dict(N=N, M=M, pose=pose, alt=alt, gr=gr, kps=k, desc=d, geo=orc.geo_at_kps(pose, gr, M, k), bb=orc.geo_bbox(pose, gr, M))
    ctx.match_pairs([0], [1])
    assert _check_pair(ctx, orc, 0, 0, 1, fr) > 20
    # L2-on-bytes branch (USE_SIFT = 1 in the reference, FEAmatcher.cpp:106-139)
    mp, op, mt, pg = ctx.default_params()
    mt.use_l2 = 1
    ctx.set_params(match=mt)
    op_ = orc.match_params(); op_.use_l2 = 1
    ctx.match_pairs([0], [1])
    _check_pair(ctx, orc, 0, 0, 1, fr, op_)
    mt.use_l2 = 0
    ctx.set_params(match=mt)


def test_matcher_disjoint_and_batch(ctx, orc):
    """a batch of pairs in one call, including frames whose geo boxes do not intersect"""
    fr = _mkframes(orc, ctx, (250, 250, 250, 40), seed=11)
    frames = {i: f for i, f in enumerate(fr)}
    src = [0, 0, 0, 1, 1, 2]; tgt = [1, 2, 3, 2, 3, 3]
    ctx.match_pairs(src, tgt)
    tot = 0
    for p, (i, j) in enumerate(zip(src, tgt)):
        tot += _check_pair(ctx, orc, p, i, j, frames)
    assert ctx.match_total()[0] == tot


def test_descriptor_distance(ctx, orc):
    fr = _mkframes(orc, ctx, (64, 64), seed=13, same_landmarks=False)
    for (ia, ib) in ((0, 0), (5, 9), (63, 1)):
        assert ctx.descriptor_distance(0, ia, 1, ib) == orc.lib().orc_hamming256(orc.u8(fr[0]["desc"][ia]), orc.u8(fr[1]["desc"][ib]))


def test_matcher_on_extracted_features(ctx, orc):
    """features from the oracle's ORB extractor on a synthetic survey: realistic candidate counts and ratios"""
    from diasss_amd.synth import Survey
    F, N, M = 3, 700, 480
    sv = Survey(F, N, M, seed=21)
    fr = {}
    for f in range(F):
        raw = sv.frame(f).numpy()
        pose, alt, gr = sv.inputs(f)
        kps, desc, _, _ = orc.detect_feature(raw)
        ctx.frame_set(f, None, N, M, pose, alt, gr)
        ctx.features_set(f, N, M, kps, desc)
        fr[f] = dict(N=N, M=M, pose=pose, alt=alt, gr=gr, kps=kps, desc=desc, geo=orc.geo_at_kps(pose, gr, M, kps), bb=orc.geo_bbox(pose, gr, M))
        assert abs(ctx.overlap(0, f) - orc.lib().orc_overlap(orc.dp(fr[0]["bb"]), orc.dp(fr[f]["bb"]))) == 0
    src = [0, 0, 1]; tgt = [1, 2, 2]
    ctx.match_pairs(src, tgt)
    n = [_check_pair(ctx, orc, p, i, j, fr) for p, (i, j) in enumerate(zip(src, tgt))]
    assert n[0] > 10 and n[2] > 10


@pytest.mark.parametrize("kind", ["lattice", "cluster", "cell_edges"])
@pytest.mark.parametrize("grid", ["1", "0"])
def test_matcher_geo_grid_corner_cases(ctx, orc, kind, grid, monkeypatch):
    """the geo grid of the matcher (and the all-pairs kernel, DSSS_MT_GRID=0) on geometry chosen against it: neighbours at exactly the
    radius (excluded: strict <) and one ulp inside it, diagonal neighbours either side of it, every keypoint in ONE cell, points on the
    cell borders and on the far edges of the box; few distinct descriptors, so that the minimum is tied all the time (lowest index wins,
    second = best).  Explicit geo points and boxes through dsss_features_set."""
    from tests import helpers as H
    monkeypatch.setenv("DSSS_MT_GRID", grid)
    N, M = 700, 480
    rng = np.random.default_rng({"lattice": 1, "cluster": 2, "cell_edges": 3}[kind])
    bb = np.array([-20.0, 100.0, 5.0, 65.0])                    # x0 x1 y0 y1
    r = 8.0
    if kind == "lattice":
        gx, gy = np.meshgrid(bb[0] + 4 + r * np.arange(14), bb[2] + 4 + r * np.arange(7))
        B = np.stack([gx.ravel(), gy.ravel()], 1)
        d = 5.65685424949238                                     # 8 / sqrt 2: the diagonal neighbour lies at the radius to within rounding
        offs = np.array([[0, 0], [r, 0], [np.nextafter(r, 0), 0], [0, -r], [0, np.nextafter(r, 0)], [d, d], [np.nextafter(d, 0), np.nextafter(d, 0)],
                         [np.nextafter(d, 9), np.nextafter(d, 9)], [-d, d], [7.9, 1.2], [1.3, -7.95]])
        A = (B[:, None, :] + offs[None, :, :]).reshape(-1, 2)
    elif kind == "cluster":
        B = np.array([30.0, 40.0]) + rng.uniform(-1e-3, 1e-3, (2000, 2))
        A = np.concatenate([np.array([30.0, 40.0]) + rng.uniform(-1e-3, 1e-3, (200, 2)), np.array([30.0, 40.0]) + rng.uniform(-9, 9, (300, 2))])
    else:
        cs = r / 2 * (1.0 + 1.0 / 1048576.0)                     # the grid's cell (dsss_match.hip: MT_SUB = 2 cells per radius)
        edges_x = bb[0] + cs * np.arange(1, 27); edges_y = bb[2] + cs * np.arange(1, 13)
        bx = np.concatenate([edges_x, np.nextafter(edges_x, -1e9), [bb[0], bb[1], bb[1]]])
        by = np.concatenate([edges_y, np.nextafter(edges_y, 1e9), [bb[2], bb[3]]])
        gx, gy = np.meshgrid(bx, by)
        B = np.stack([gx.ravel(), gy.ravel()], 1)
        offs = np.array([[0, 0], [np.nextafter(r, 0), 0], [-np.nextafter(r, 0), 0], [0, np.nextafter(r, 0)], [0, -np.nextafter(r, 0)], [r, 0], [-r, 0], [3, 3]])
        A = (B[::6, None, :] + offs[None, :, :]).reshape(-1, 2)
    A = A[(A[:, 0] >= bb[0] - 3) & (A[:, 0] <= bb[1] + 3) & (A[:, 1] >= bb[2] - 3) & (A[:, 1] <= bb[3] + 3)]       # a few just outside the box: skipped (FEAmatcher.cpp:84)
    assert len(A) <= 2000 and len(B) <= 2000
    palette = rng.integers(0, 256, (4, 32), dtype=np.uint8)
    fr = {}
    for f, G in ((0, A), (2, B)):
        n = len(G)
        pose, alt, gr = H.track(N, M, 0, seed=3)
        kps, _ = H.random_features(N, M, n, 7 + f)
        desc = palette[rng.integers(0, 4, n)].copy()
        flip = rng.integers(0, 3, n)                              # 0, 1 or 2 flipped bits: distances 0 .. 4 inside a palette entry
        for i in range(n):
            for b in rng.choice(256, flip[i], replace=False):
                desc[i, b // 8] ^= np.uint8(1 << (b % 8))
        ctx.frame_set(f, None, N, M, pose, alt, gr)
        ctx.features_set(f, N, M, kps, desc, geo=G, bbox=bb)
        fr[f] = dict(N=N, M=M, pose=pose, alt=alt, gr=gr, kps=kps, desc=desc, geo=np.ascontiguousarray(G), bb=bb)
    ctx.match_pairs([0], [2])
    _check_pair(ctx, orc, 0, 0, 2, fr)
    nn = ctx.match_dir(0, 0)[0][:len(A)]
    assert (nn >= 0).sum() > 10


@pytest.mark.parametrize("radius", [0.004, 0.5, 8.0, 50.0, 1000.0])
def test_matcher_radius_sweep(ctx, orc, radius):
    """the search radius from far below the keypoint spacing (the geo box would hold more than 2^22 cells: the call falls back to the
    all-pairs kernel) to far above the box (one cell): geo grid against the oracle, and against DSSS_MT_GRID=0, on the features of a
    synthetic pair"""
    import os
    fr = _mkframes(orc, ctx, (600, 600), seed=17)
    frames = {0: fr[0], 1: fr[1]}
    mp, op, mt, pg = ctx.default_params()
    mt.radius = radius
    ctx.set_params(match=mt)
    op_ = orc.match_params(); op_.radius = radius
    try:
        ctx.match_pairs([0], [1])
        _check_pair(ctx, orc, 0, 0, 1, frames, op_)
        nn_grid = [ctx.match_dir(0, d)[0].copy() for d in (0, 1)]
        os.environ["DSSS_MT_GRID"] = "0"
        ctx.match_pairs([0], [1])
        for d in (0, 1):
            assert (ctx.match_dir(0, d)[0] == nn_grid[d]).all()
    finally:
        os.environ.pop("DSSS_MT_GRID", None)
        mt.radius = 8.0
        ctx.set_params(match=mt)
