"""CPU: known-answer tests of the oracle's 128-element descriptor (oracle/orc_sift.c; SURVEY.md section 8f, N4 -- the SIFT call
site of /root/reference/thirdparty/ORBextractor.cpp:1043-1047,1098 as intended, matched by the L2 branch of
/root/reference/src/core/FEAmatcher.cpp:106-139).  OpenCV is not in the image: these pin the restatement from first principles."""
import numpy as np


def _ramp(n=200, slope=1):
    return np.tile(np.clip(np.arange(n) * slope, 0, 255).astype(np.uint8), (n, 1))


def test_constant_gradient_patch_fills_one_orientation_bin(orc):
    """dx = 2, dy = 0 everywhere: orientation 0 relative to a keypoint angle of 0 -> bin 0 of every spatial cell and nothing else;
    the Gaussian window makes the four centre cells the heaviest and the pattern symmetric"""
    h = orc.sift_hist(_ramp(), 100, 100, 0.0).reshape(4, 4, 8)
    assert (h[..., 1:] == 0).all() and (h[..., 0] > 0).all()
    assert (h[..., 0] == h[::-1, :, 0]).all() and (h[..., 0] == h[:, ::-1, 0]).all() and (h[..., 0] == h[..., 0].T).all()
    assert h[1, 1, 0] > h[0, 1, 0] > h[0, 0, 0]
    d = orc.sift128(_ramp(), 100, 100, 0.0).reshape(4, 4, 8)
    assert (d[..., 1:] == 0).all()
    # 16 equal-ish entries: none is clipped (0.2 x norm > each), so the row has norm 512 up to the rounding of its elements
    assert abs(np.sqrt((d.astype(np.float64) ** 2).sum()) - 512) < 2


def test_keypoint_angle_shifts_the_orientation_bin(orc):
    """ori = 360 - kpt.angle (cv::SIFT::compute); a keypoint angle of 45 k degrees moves the whole mass to bin k"""
    for k in range(8):
        d = orc.sift128(_ramp(), 100, 100, 45.0 * k).reshape(16, 8).sum(0)
        assert d[k] > 0 and d.sum() == d[k], (k, d)


def test_rotation_covariance(orc):
    """the image turned by a quarter turn and the keypoint angle with it: the SAME 128 numbers (a quarter turn maps the sample grid onto
    itself; sin / cos of the two angles differ in the last bit at most, which moves no rounded share here)"""
    rng = np.random.default_rng(1)
    im = orc.blur13(rng.integers(0, 256, (201, 201)).astype(np.uint8))
    for ang in (30.0, 171.5, 300.25):
        d0 = orc.sift128(im, 100, 100, ang)
        d1 = orc.sift128(np.ascontiguousarray(np.rot90(im, 1)), 100, 100, (ang - 90.0) % 360.0)
        assert np.abs(d0.astype(int) - d1.astype(int)).max() <= 1 and (d0 != d1).sum() <= 2
        assert d0.max() > 0
        d2 = orc.sift128(im, 100, 100, (ang + 45.0) % 360.0)
        assert (d0 != d2).sum() > 64                                   # (another angle IS another descriptor)


def test_clip_and_renormalise_rule(orc):
    """normalise, clip at 0.2, renormalise, x 512, saturate (Lowe; OpenCV's calcSIFTDescriptor tail) on exact integers"""
    one = np.zeros(128, np.int32); one[0] = 1000
    assert orc.sift_finalize(one)[0] == 255 and (orc.sift_finalize(one)[1:] == 0).all()       # 200 / 200 x 512 saturates
    flat = orc.sift_finalize(np.full(128, 777, np.int32))
    assert (flat == 45).all()                                          # 512 / sqrt(128) = 45.25: nothing clipped
    assert (orc.sift_finalize(np.zeros(128, np.int32)) == 0).all()     # an empty window is a zero row, not NaN
    h = np.full(128, 100, np.int32); h[5] = 100000                     # one dominant entry: clipped to 0.2 x norm = 20 000
    out = orc.sift_finalize(h).astype(np.float64)
    want = np.minimum(h, int(0.2 * np.sqrt((h.astype(np.float64) ** 2).sum())))
    want = np.clip(np.rint(want * 512.0 / np.sqrt((want.astype(np.float64) ** 2).sum())), 0, 255)
    assert (out == want).all() and out[5] == 255 and out[0] == 3
    # scale invariance of the fixed point: 16 x the histogram is the same row
    rng = np.random.default_rng(2)
    h = rng.integers(0, 5000, 128).astype(np.int32)
    assert (orc.sift_finalize(h) == orc.sift_finalize(h * 16)).sum() >= 126


def test_gaussian_window_table(orc):
    w = orc.sift_weights(1569)
    k = np.arange(1569)
    assert w[0] == 1.0 and np.abs(w - np.exp(-k / 512.0)).max() < 1e-7 and (np.diff(w) < 0).all()


def test_window_clipped_at_the_image_border(orc):
    """samples with r <= 0, r >= rows - 1, c <= 0 or c >= cols - 1 are skipped (calcSIFTDescriptor): a keypoint 19 px from the corner
    (the extractor's EDGE_THRESHOLD) still gets a descriptor, from fewer samples"""
    im = _ramp(120)                                                     # the same gradient everywhere: the mass counts the samples
    hc = orc.sift_hist(im, 60, 60, 10.0); hb = orc.sift_hist(im, 19, 19, 10.0); hn = orc.sift_hist(im, 5, 60, 10.0)
    assert 0 < hn.sum() < hb.sum() < hc.sum()
    assert orc.sift128(im, 19, 19, 10.0).max() > 0


def test_extractor_emits_both_descriptors_and_l2_matching_uses_the_128_rows(orc):
    """orc_orb_extract_sift = orc_orb_extract + the 128 rows at the same keypoints; the matcher's L2 branch on them (use_l2 = 2)"""
    rng = np.random.default_rng(5)
    yy, xx = np.mgrid[0:300, 0:260]
    img = np.zeros((300, 260))
    for _ in range(150):
        cx, cy, s, a = rng.uniform(0, 260), rng.uniform(0, 300), rng.uniform(2, 6), rng.uniform(40, 200)
        img += a * np.exp(-((xx - cx) ** 2 + (yy - cy) ** 2) / (2 * s * s))
    img = np.clip(img + rng.normal(0, 4, img.shape), 0, 255).astype(np.uint8)
    k0, d0 = orc.orb_extract(img)
    k1, d1, s1 = orc.orb_extract_sift(img)
    assert len(k0) == len(k1) > 100 and (d0 == d1).all() and (k0 == k1).all()
    assert s1.shape == (len(k1), 128) and (s1.max(1) > 0).all()
    nrm = np.sqrt((s1.astype(np.float64) ** 2).sum(1))
    assert np.abs(nrm - 512).max() < 40                                 # unit rows x 512, up to saturation and rounding
    # one keypoint checked by hand: level image, blur, integer level coordinates, IC angle
    i = int(np.argmax(k1["octave"] == 0))
    want = orc.sift128(orc.blur13(img), int(round(k1["x"][i])), int(round(k1["y"][i])), float(k1["angle"][i]))
    assert (want == s1[i]).all()
    # matching a frame against itself with the 128 rows: every keypoint inside the box finds itself at distance 0
    geo = np.stack([k1["x"], k1["y"]], 1).astype(np.float64) * 0.05
    bb = np.array([geo[:, 0].min(), geo[:, 0].max(), geo[:, 1].min(), geo[:, 1].max()])
    p = orc.match_params(); p.use_l2 = 2
    r = orc.match_dir(0, 2, 300, k1, s1, geo, k1, s1, geo, bb, p, scc=False)
    assert (r["best"] == 0).all()
    hit = r["nn"] >= 0
    assert hit.sum() > 0.3 * len(k1) and (r["nn"][hit] == np.arange(len(k1))[hit]).all()
