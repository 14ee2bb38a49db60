"""GPU parity: preprocessing + ORB extraction kernels against the CPU oracle, stage by stage, bit-exact."""
import ctypes as C
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from diasss_amd import capi
    c = capi.Context(max_frames=4)
    yield c
    c.close()


def _frame(N, M, seed, hot=True):
    from diasss_amd.synth import Survey
    sv = Survey(2, N, M, seed=seed)
    raw = sv.frame(1).numpy().copy()
    if hot:   # a few "sensor buggy line" pixels inside the valid area (frame.cpp:98-103)
        rng = np.random.default_rng(seed)
        for _ in range(5):
            raw[rng.integers(160, N - 160), rng.integers(100, M - 100)] = 4.0 * raw.mean()
        raw[200, 150] = 3.0 * raw.mean()
    pose, alt, gr = sv.inputs(1)
    return raw, pose, alt, gr


@pytest.mark.parametrize("N,M,seed", [(640, 400, 3), (500, 700, 4), (1000, 512, 5)])
def test_extract_stage_parity(ctx, orc, N, M, seed):
    raw, pose, alt, gr = _frame(N, M, seed)
    ctx.frame_set(0, raw, N, M, pose, alt, gr)
    n = ctx.extract(0)
    # K1: normalised image + mask
    norm, mask = ctx.frame_norm(0, N, M)
    o_norm = orc.normalize(raw); o_mask = orc.mask(raw)
    assert (norm == o_norm).all(), "normalised image differs"
    assert (mask == o_mask).all(), "filter mask differs"
    # K2: pyramid
    p = orc.orb_params()
    lr = (C.c_int * 8)(); lc = (C.c_int * 8)()
    orc.lib().orc_orb_level_sizes(N, M, C.byref(p), lr, lc)
    prev = o_norm
    levels = [o_norm]
    for l in range(1, 6):
        dst = np.zeros((lr[l], lc[l]), np.uint8)
        orc.lib().orc_resize_linear_u8(orc.u8(prev), prev.shape[0], prev.shape[1], orc.u8(dst), lr[l], lc[l])
        g = ctx.frame_level(0, l, lr[l], lc[l])
        assert g.shape == dst.shape and (g == dst).all(), "pyramid level %d differs" % l
        prev = dst; levels.append(dst)
    # K3: FAST candidates per level, reference order
    for l in range(6):
        cap = 200000
        xs = np.zeros(cap, np.float32); ys = np.zeros(cap, np.float32); rs = np.zeros(cap, np.float32)
        k = orc.lib().orc_fast_level(orc.u8(levels[l]), lr[l], lc[l], 12, 7, orc.fp(xs), orc.fp(ys), orc.fp(rs), cap)
        gx, gy, gr_ = ctx.frame_candidates(0, l)
        assert len(gx) == k, "level %d: %d candidates vs oracle %d" % (l, len(gx), k)
        assert (gx == xs[:k]).all() and (gy == ys[:k]).all() and (gr_ == rs[:k]).all()
    # K4-K7 + mask filter: final features
    kps, desc, norm2, mask2 = orc.detect_feature(raw)
    g_kps, g_desc, g_geo = ctx.features_get(0)
    assert n == len(kps) == len(g_kps)
    for fld in ("x", "y", "size", "angle", "response", "octave"):
        assert (g_kps[fld] == kps[fld]).all(), "keypoint field %s differs" % fld
    assert (g_desc == desc).all(), "descriptors differ"
    assert (g_geo == orc.geo_at_kps(pose, gr, M, kps)).all()
    assert n > 200


def test_extract_small_orb_config_and_reuse(ctx, orc):
    """non-default ORB parameters (nfeatures 500, 4 levels) and context reuse with a different frame size"""
    N, M = 640, 400
    raw, pose, alt, gr = _frame(N, M, 9, hot=False)
    mp, op, mt, pg = ctx.default_params()
    op.nfeatures = 500; op.nlevels = 4
    ctx.set_params(orb=op)
    ctx.frame_set(1, raw, N, M, pose, alt, gr)
    n = ctx.extract(1)
    po = orc.orb_params(); po.nfeatures = 500; po.nlevels = 4
    kps, desc, _, _ = orc.detect_feature(raw, None, po)
    g_kps, g_desc, _ = ctx.features_get(1)
    assert n == len(kps) and (g_desc == desc).all() and (g_kps["x"] == kps["x"]).all() and (g_kps["angle"] == kps["angle"]).all()
    dp = ctx.default_params()[1]
    ctx.set_params(orb=dp)


@pytest.mark.parametrize("scale,N,M", [(2.0, 1000, 802), (1.5, 801, 598), (1.1, 501, 334)])
def test_extract_other_scale_factors(ctx, orc, scale, N, M):
    """pyramid scale factors other than 1.2 and odd sizes: at 2.0 the source bytes of four output pixels do not fit the
    12-byte windows of resize_kernel and the pixel-by-pixel form runs; every level must match the oracle's byte for byte"""
    raw, pose, alt, gr = _frame(N, M, 11, hot=False)
    mp, op, mt, pg = ctx.default_params()
    op.nfeatures = 600; op.nlevels = 4; op.scale = scale
    ctx.set_params(orb=op)
    try:
        ctx.frame_set(2, raw, N, M, pose, alt, gr)
        n = ctx.extract(2)
        po = orc.orb_params(); po.nfeatures = 600; po.nlevels = 4; po.scale = scale
        kps, desc, _, _ = orc.detect_feature(raw, None, po)
        g_kps, g_desc, _ = ctx.features_get(2)
        assert n == len(kps) and n > 50
        for fld in ("x", "y", "octave", "angle", "response"):
            assert (g_kps[fld] == kps[fld]).all(), fld
        assert (g_desc == desc).all()
    finally:
        ctx.set_params(orb=ctx.default_params()[1])


def test_resize_strips_and_flat_form_in_one_batch(ctx, orc):
    """at a scale of 1.334 the 12-byte windows fit the 500 x 300 frame's first level (resize_strip_kernel) and do not fit the
    500 x 302 frame's (resize_kernel, which then skips the frames the strips did): both in ONE extract_many, every keypoint and
    descriptor as the oracle's"""
    mp, op, mt, pg = ctx.default_params()
    op.nfeatures = 500; op.nlevels = 4; op.scale = 1.334
    ctx.set_params(orb=op)
    try:
        frames = [_frame(500, M, 21 + i, hot=False) for i, M in enumerate((300, 302))]
        for i, (raw, pose, alt, gr) in enumerate(frames): ctx.frame_set(i, raw, 500, raw.shape[1], pose, alt, gr)
        ctx.extract_many([0, 1])
        po = orc.orb_params(); po.nfeatures = 500; po.nlevels = 4; po.scale = 1.334
        for i, (raw, pose, alt, gr) in enumerate(frames):
            kps, desc, _, _ = orc.detect_feature(raw, None, po)
            g_kps, g_desc, _ = ctx.features_get(i)
            assert len(g_kps) == len(kps) and len(kps) > 50
            for fld in ("x", "y", "octave", "angle", "response"):
                assert (g_kps[fld] == kps[fld]).all(), (i, fld)
            assert (g_desc == desc).all()
    finally:
        ctx.set_params(orb=ctx.default_params()[1])


def test_extract_device_resident_input(ctx, orc):
    """raw image handed over as a device pointer (torch tensor in HBM): same result, no host copy"""
    import torch
    N, M = 640, 400
    raw, pose, alt, gr = _frame(N, M, 12, hot=False)
    t = torch.from_numpy(raw).cuda()
    ctx.frame_set(2, t, N, M, pose, alt, gr)
    n = ctx.extract(2)
    kps, desc, _, _ = orc.detect_feature(raw)
    g_kps, g_desc, _ = ctx.features_get(2)
    assert n == len(kps) and (g_desc == desc).all()


def test_eager_extraction_started_by_frames_set_and_its_fallbacks(ctx, orc):
    """dsss_frames_set starts the extraction of device-resident frames (as the reference's Frame constructor runs DetectFeature);
    dsss_extract_many over the same list only finishes it.  Whatever happens in between -- another list of frames, parameters changed,
    frames set again -- the features must be those of the oracle for the parameters in force at the extraction."""
    import torch
    N, M = 640, 400
    frames = [_frame(N, M, 20 + k, hot=(k == 0)) for k in range(3)]
    dev = [torch.from_numpy(f[0]).cuda() for f in frames]
    ids = [0, 1, 2]

    def set_all():
        ctx.frames_set(ids, dev, [N] * 3, [M] * 3, [f[1] for f in frames], [f[2] for f in frames], [f[3] for f in frames])

    def check(po=None):
        for k in ids:
            kps, desc, _, _ = orc.detect_feature(frames[k][0], None, po) if po is not None else orc.detect_feature(frames[k][0])
            g_kps, g_desc, g_geo = ctx.features_get(k)
            assert len(g_kps) == len(kps) and (g_desc == desc).all() and (g_kps["x"] == kps["x"]).all() and (g_kps["angle"] == kps["angle"]).all()
            ref_geo = orc.geo_at_kps(frames[k][1], frames[k][3], M, kps)
            assert (np.asarray(g_geo).reshape(-1, 2) == np.asarray(ref_geo).reshape(-1, 2)).all()       # the one kernel that waits for the geometry upload

    set_all(); ctx.extract_many(ids); check()                                   # the plain pair: only the tail runs in extract_many
    set_all(); ctx.extract_many([2, 0, 1]); check()                             # another order: the started extraction is dropped, all of it runs again
    set_all(); set_all(); ctx.extract_many(ids); check()                        # set twice: the second start is the one that counts
    mp, op, mt, pg = ctx.default_params()
    op.nfeatures = 500; op.nlevels = 4
    try:
        set_all(); ctx.set_params(orb=op); ctx.extract_many(ids)                # parameters changed after the start: extracted with the NEW ones
        po = orc.orb_params(); po.nfeatures = 500; po.nlevels = 4
        check(po)
    finally:
        ctx.set_params(orb=ctx.default_params()[1])
    set_all(); assert ctx.extract(1) > 0; ctx.extract_many(ids); check()        # a single-frame extraction in between


def test_full_geo_image_on_request(ctx, orc):
    """Frame::geo_img as the reference holds it (frame.cpp:126-165: the N x M pair) through dsss_frame_get_geo -- the hot path only
    ever uses its extremes (dsss_frame_bbox) and its samples at the keypoints; a caller that reads the field gets the full image,
    bit for bit the oracle's, and Util::ComputeIntersection sees the same extremes on either form"""
    from tests import helpers as H
    for leg, (N, M) in enumerate(((700, 480), (333, 130))):
        pose, alt, gr = H.track(N, M, leg, seed=9)
        raw = np.random.default_rng(leg).rayleigh(1.0, (N, M)) * 100.0
        ctx.frame_set(leg, raw, N, M, pose, alt, gr)
        gx, gy = ctx.frame_geo(leg, N, M)
        ox, oy = orc.geo_img(pose, gr, M)
        assert (gx == ox).all() and (gy == oy).all()
        bb = ctx.frame_bbox(leg)
        assert bb[0] == gx.min() and bb[1] == gx.max() and bb[2] == gy.min() and bb[3] == gy.max()
