"""First-principles known-answer tests that pin the CPU oracle.

The reference has no tests, fixtures or data and cannot be built here (OpenCV/GTSAM absent), so these
hand-derived answers are what anchors the oracle: "parity unpinned" at the OpenCV/GTSAM boundary.
"""
import ctypes as C
import math
import numpy as np
import pytest


# (1) DescriptorDistance SWAR == popcount (FEAmatcher.cpp:442-458)
def test_hamming_swar_equals_popcount(orc):
    rng = np.random.default_rng(1)
    L = orc.lib()
    for _ in range(200):
        a = rng.integers(0, 256, 32, dtype=np.uint8); b = rng.integers(0, 256, 32, dtype=np.uint8)
        ref = int(np.unpackbits(a ^ b).sum())
        assert L.orc_hamming256(orc.u8(a), orc.u8(b)) == ref
    a = rng.integers(0, 256, 32, dtype=np.uint8)
    assert L.orc_hamming256(orc.u8(a), orc.u8(a)) == 0
    assert L.orc_hamming256(orc.u8(a), orc.u8(~a)) == 256


def _one_vs_many(orc, dists, use_l2=False):
    """kp a at the origin, candidates all inside the gate with prescribed Hamming distances"""
    n = len(dists)
    kpa = np.zeros(1, orc.KP_DTYPE); kpb = np.zeros(n, orc.KP_DTYPE)
    da = np.zeros((1, 32), np.uint8); db = np.zeros((n, 32), np.uint8)
    for j, d in enumerate(dists):
        bits = np.zeros(256, np.uint8); bits[:d] = 1
        db[j] = np.packbits(bits)
    ga = np.zeros((1, 2)); gb = np.zeros((n, 2)); gb[:, 0] = 0.5
    bb = np.array([-10., 10., -10., 10.])
    return orc.match_dir(0, 2, 100, kpa, da, ga, kpb, db, gb, bb, scc=False)


# (2) best / second-best / tie semantics (FEAmatcher.cpp:148-175)
def test_best_second_tie_semantics(orc):
    r = _one_vs_many(orc, [5, 5]); assert (r["best"][0], r["second"][0]) == (5, 5) and r["nn"][0] == -1  # ratio 1 > 0.35
    r = _one_vs_many(orc, [7, 5]); assert (r["best"][0], r["second"][0]) == (5, 7)
    r = _one_vs_many(orc, [10, 40]); assert r["nn"][0] == 0          # 10/40 = 0.25 <= 0.35
    r = _one_vs_many(orc, [40, 10, 10]); assert r["nn"][0] == -1     # tie on best: first index keeps it, ratio 1
    r = _one_vs_many(orc, [10, 100, 10]); assert r["nn"][0] == -1 and r["second"][0] == 10
    r = _one_vs_many(orc, [88]); assert r["nn"][0] == 0              # single candidate, best <= 88
    r = _one_vs_many(orc, [89]); assert r["nn"][0] == -1
    r = _one_vs_many(orc, [14, 40]); assert r["nn"][0] == 0          # 0.35 exactly: 14/40
    r = _one_vs_many(orc, [15, 40]); assert r["nn"][0] == -1
    r = _one_vs_many(orc, [30, 86, 90]); assert r["nn"][0] == 0      # 30/86 = 0.3488


def test_gate_radius_and_bbox(orc):
    kpa = np.zeros(1, orc.KP_DTYPE); kpb = np.zeros(2, orc.KP_DTYPE)
    da = np.zeros((1, 32), np.uint8); db = np.zeros((2, 32), np.uint8)
    ga = np.zeros((1, 2)); gb = np.array([[8.0, 0.0], [7.999999, 0.0]])
    bb = np.array([-10., 10., -10., 10.])
    r = orc.match_dir(0, 2, 100, kpa, da, ga, kpb, db, gb, bb, scc=False)
    assert r["ncand"][0] == 1 and r["nn"][0] == 1      # dist < 8 strict
    r = orc.match_dir(0, 2, 100, kpa, da, ga + 11.0, kpb, db, gb + 11.0, bb, scc=False)
    assert r["ncand"][0] == 0                          # outside the reference bbox: skipped
    # sqrt(d2) < 8  <=>  d2 < 64 for correctly rounded sqrt (the HIP gate compares d2 with a host-computed
    # threshold T = min{d : sqrt(d) >= radius} instead of taking the sqrt)
    assert math.sqrt(np.nextafter(64.0, 0.0)) < 8.0 and math.sqrt(64.0) == 8.0


# (3) cv::RNG MWC stream from the default state
def test_cv_rng_stream(orc):
    st = C.c_uint64(0xFFFFFFFF)
    got = [orc.lib().orc_cvrng_next(C.byref(st)) for _ in range(6)]
    # recompute from the published recurrence in Python ints
    s = 0xFFFFFFFF; exp = []
    for _ in range(6):
        s = (s & 0xFFFFFFFF) * 4164903690 + (s >> 32); s &= (1 << 64) - 1; exp.append(s & 0xFFFFFFFF)
    assert got == exp
    assert got == [130063606, 3003295397, 3870020839, 1350273629, 4024955497, 3216027310]


# (4) libstdc++ default_random_engine + normal_distribution(0,1): first 12 draws captured with g++ 11.4
def test_libstdcxx_normal_stream(orc):
    z = np.zeros(12); orc.lib().orc_normal_fill(orc.dp(z), 12)
    exp = [-0.12196578414159691, -1.0868180442613573, 0.68428994379655483, -1.075189149518029,
           0.03326947642049239, 0.74483559772278241, 0.03360612264682257, -0.52663720618529819,
           0.46253204358022892, 0.20069944199703771, 2.1246766763318949, 0.37172123056217998]
    assert np.allclose(z, exp, rtol=0, atol=1e-15)


# (5) per-level quotas and pyramid sizes (ORBextractor.cpp:435-446, 1119-1120)
def test_level_quota_and_sizes(orc):
    p = orc.orb_params()
    q = (C.c_int * 8)(); orc.lib().orc_orb_level_quota(C.byref(p), q)
    assert list(q)[:6] == [501, 418, 348, 290, 242, 201]
    r = (C.c_int * 8)(); c = (C.c_int * 8)()
    orc.lib().orc_orb_level_sizes(2000, 1024, C.byref(p), r, c)
    assert list(zip(list(r)[:6], list(c)[:6])) == [(2000, 1024), (1667, 853), (1389, 711), (1157, 593), (965, 494), (804, 412)]


# (6) umax table (ORBextractor.cpp:454-469)
def test_umax(orc):
    u = (C.c_int * 17)(); orc.lib().orc_orb_umax(u)
    assert list(u)[:16] == [15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3]


# (7) FAST on hand-made images
def _fast(orc, img, thr):
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape
    xs = np.zeros(1024, np.int32); ys = np.zeros(1024, np.int32); sc = np.zeros(1024, np.int32)
    n = orc.lib().orc_fast_window(orc.u8(img), w, h, w, thr, orc.ip(xs), orc.ip(ys), orc.ip(sc), 1024)
    return [(int(xs[i]), int(ys[i]), int(sc[i])) for i in range(n)]


def test_fast_single_bright_pixel(orc):
    img = np.full((15, 15), 50, np.uint8); img[7, 7] = 200
    # the centre is darker-ring corner: all 16 ring pixels are 150 below it -> A = 150, score 149
    assert _fast(orc, img, 12) == [(7, 7, 149)]
    assert _fast(orc, img, 150) == []                       # needs A > thr


def test_fast_step_edge_is_not_a_corner(orc):
    img = np.full((15, 15), 50, np.uint8); img[:, 8:] = 200   # straight edge: at most 7-8 contiguous -> no corner
    assert _fast(orc, img, 12) == []


def test_fast_quadrant_corner_and_nms(orc):
    img = np.full((21, 21), 40, np.uint8); img[10:, 10:] = 220  # bright quadrant: 11 contiguous dark ring pixels at its tip
    assert _fast(orc, img, 12) == []      # (10,10) and (11,11) tie at A = 180: strict NMS removes both
    img[10, 10] = 230                     # break the tie
    got = _fast(orc, img, 12)
    assert (10, 10, 189) in got
    xs = [g[0] for g in got]; ys = [g[1] for g in got]
    assert all(8 <= x <= 12 and 8 <= y <= 12 for x, y in zip(xs, ys))
    # NMS: no two detections are 8-neighbours
    for i in range(len(got)):
        for j in range(i + 1, len(got)):
            assert max(abs(got[i][0] - got[j][0]), abs(got[i][1] - got[j][1])) > 1
    # brute-force definition check of the arc value at the tip
    A = np.zeros((21, 21), np.int32); orc.lib().orc_fast_arc_map(orc.u8(img), 21, 21, 21, orc.ip(A))
    ring = [(0, 3), (1, 3), (2, 2), (3, 1), (3, 0), (3, -1), (2, -2), (1, -3), (0, -3), (-1, -3), (-2, -2), (-3, -1), (-3, 0), (-3, 1), (-2, 2), (-1, 3)]
    for (x, y) in [(10, 10), (9, 9), (11, 11), (5, 5)]:
        v = int(img[y, x]); d = [v - int(img[y + dy, x + dx]) for dx, dy in ring]
        best = 0
        for k in range(16):
            arc = [d[(k + j) % 16] for j in range(9)]
            best = max(best, min(arc), -max(arc))
        assert A[y, x] == best


def test_fast_window_border_rule(orc):
    img = np.full((9, 9), 50, np.uint8); img[3, 3] = 250   # first evaluable pixel
    assert _fast(orc, img, 12) == [(3, 3, 199)]
    img = np.full((9, 9), 50, np.uint8); img[2, 3] = 250   # row 2 is outside [3, h-3): ring pixel only
    assert all(g[1] >= 3 for g in _fast(orc, img, 12))


# fastAtan2 accuracy (OpenCV documents ~0.3 deg) and quadrant handling
def test_fast_atan2(orc):
    f = orc.lib().orc_fast_atan2
    for y, x in [(0, 1), (1, 1), (1, 0), (1, -1), (0, -1), (-1, -1), (-1, 0), (-1, 1), (3, 7), (-120, 33)]:
        ref = math.degrees(math.atan2(y, x)) % 360.0
        assert abs(f(float(y), float(x)) - ref) < 0.3
    assert f(0.0, 0.0) == 0.0


# deterministic sincos vs libm
def test_sincos_det(orc):
    s = C.c_double(); c = C.c_double()
    for a in np.linspace(-40, 40, 4001):
        orc.lib().orc_sincos(float(a), C.byref(s), C.byref(c))
        assert abs(s.value - math.sin(a)) < 4e-16 and abs(c.value - math.cos(a)) < 4e-16


# resize: constant image stays constant; identity-size copy; known 2:1 box case
def test_resize_linear(orc):
    src = np.full((60, 48), 77, np.uint8); dst = np.zeros((50, 40), np.uint8)
    orc.lib().orc_resize_linear_u8(orc.u8(src), 60, 48, orc.u8(dst), 50, 40)
    assert (dst == 77).all()
    rng = np.random.default_rng(3)
    src = rng.integers(0, 256, (33, 29), dtype=np.uint8); dst = np.zeros_like(src)
    orc.lib().orc_resize_linear_u8(orc.u8(src), 33, 29, orc.u8(dst), 33, 29)
    assert (dst == src).all()
    src = rng.integers(0, 256, (40, 40), dtype=np.uint8); dst = np.zeros((20, 20), np.uint8)
    orc.lib().orc_resize_linear_u8(orc.u8(src), 40, 40, orc.u8(dst), 20, 20)
    box = (src[0::2, 0::2].astype(int) + src[0::2, 1::2] + src[1::2, 0::2] + src[1::2, 1::2] + 2) >> 2
    assert (dst == box).all()     # fx = fy = 0.5 exactly: (a+b+c+d+2)>>2


def test_gauss_taps_and_blur(orc):
    t = (C.c_int * 13)(); orc.lib().orc_gauss13_taps(t)
    t = list(t)
    assert sum(t) == 256 and t == t[::-1] and t[6] == max(t)
    assert t == [1, 2, 7, 16, 31, 45, 52, 45, 31, 16, 7, 2, 1]      # OpenCV's error-diffusion rule (getGaussianKernelFixedPoint_ED)
    src = np.full((40, 30), 123, np.uint8); dst = np.zeros_like(src)
    orc.lib().orc_blur13(orc.u8(src), 40, 30, orc.u8(dst))
    assert (dst == 123).all()


# IC_Angle: intensity ramp along +x gives angle 0, along +y gives 90 (image y down)
def test_ic_angle(orc):
    yy, xx = np.mgrid[0:41, 0:41]
    img = np.ascontiguousarray((xx * 4 + 20).astype(np.uint8))
    assert abs(orc.lib().orc_ic_angle(orc.u8(img), 41, 20, 20) - 0.0) < 1e-3
    img = np.ascontiguousarray((yy * 4 + 20).astype(np.uint8))
    assert abs(orc.lib().orc_ic_angle(orc.u8(img), 41, 20, 20) - 90.0) < 0.3
    img = np.ascontiguousarray(((40 - xx) * 4 + 20).astype(np.uint8))
    assert abs(orc.lib().orc_ic_angle(orc.u8(img), 41, 20, 20) - 180.0) < 0.3


# rBRIEF: rotating the image by 90 deg and the angle by 90 deg gives the same descriptor
def test_brief_rotation_consistency(orc):
    rng = np.random.default_rng(5)
    img = np.ascontiguousarray(rng.integers(0, 256, (61, 61), dtype=np.uint8))
    d0 = np.zeros(32, np.uint8); d1 = np.zeros(32, np.uint8)
    orc.lib().orc_brief(orc.u8(img), 61, 30, 30, C.c_float(0.0), orc.u8(d0))
    # np.rot90(k=-1) maps (y, x) -> (x, 60 - y): a clockwise turn in image coordinates == +90 deg (y down)
    rot = np.ascontiguousarray(np.rot90(img, k=-1))
    orc.lib().orc_brief(orc.u8(rot), 61, 30, 30, C.c_float(90.0), orc.u8(d1))
    assert (d0 == d1).all()
    assert 60 < int(np.unpackbits(d0).sum()) < 200


# (11) slant range and nadir rejection (optimizer.cpp:602-619)
def test_reprojection_known_answers(orc):
    M = 200; half = M // 2
    gr = np.linspace(0, 9.9, half); gr[40] = 4.0
    alt = np.full(50, 3.0)
    rows = np.array([[0, 1, 10.7, half + 40.9, 20.2, half - 40.2],     # int truncation: bins 140 / 59 -> idx 40 / 41
                     [0, 1, 10, half + 19, 20, half + 40],             # nadir reject at |bin - M/2| = 19
                     [0, 1, 10, half + 20, 20, half - 20],             # kept at 20
                     [0, 2, 10, half + 40, 20, half + 40]], float)     # other target id
    kp7 = orc.get_kps_pairs(rows, 1, alt, gr, alt, gr)
    assert len(kp7) == 2
    assert kp7[0, 0] == 10 and kp7[0, 1] == half + 40 and kp7[0, 3] == 20 and kp7[0, 4] == half - 41
    assert kp7[0, 2] == 5.0                                           # alt 3, gr 4 -> 5
    assert abs(kp7[0, 5] - math.hypot(3.0, gr[41])) < 1e-15 and kp7[0, 6] == 0
    assert kp7[1, 1] == half + 20


# (12) geo mapping (frame.cpp:126-165)
def test_geo_mapping(orc):
    N, M = 4, 8; half = M // 2
    gr = np.array([1.0, 2.0, 3.0, 4.0])
    pose = np.zeros((N, 6)); pose[:, 3] = 100.0; pose[:, 4] = 200.0     # yaw 0
    x = C.c_double(); y = C.c_double()
    def at(r, c):
        orc.lib().orc_geo_at(orc.dp(pose), orc.dp(gr), N, M, r, c, C.byref(x), C.byref(y)); return x.value, y.value
    PI = 3.14159265359
    gx, gy = at(0, half)          # starboard first bin: gr[0] at yaw + PI/2
    assert abs(gx - (100 + 1.0 * math.cos(PI / 2))) < 1e-12 and abs(gy - (200 + 1.0 * math.sin(PI / 2))) < 1e-12
    gx, gy = at(0, 1)             # port col 1: gr[M/2-1] at yaw - PI/2
    assert abs(gy - (200 + 4.0 * math.sin(-PI / 2))) < 1e-12
    gx, gy = at(0, half - 1)      # port innermost: gr[1]
    assert abs(gy - (200 - 2.0)) < 1e-9
    gx0, gy0 = at(0, 0)           # col 0 would read gr[M/2] (one past the end): clamped to gr[M/2-1]
    assert abs(gy0 - (200 - 4.0)) < 1e-9
    bb = orc.geo_bbox(pose, gr, M)
    assert abs(bb[2] - 196.0) < 1e-9 and abs(bb[3] - 204.0) < 1e-9


def test_overlap_iou(orc):
    a = np.array([0., 10., 0., 10.]); b = np.array([5., 15., 0., 10.])
    assert abs(orc.lib().orc_overlap(orc.dp(a), orc.dp(b)) - 50.0 / 150.0) < 1e-6
    c = np.array([20., 30., 0., 10.])
    assert orc.lib().orc_overlap(orc.dp(a), orc.dp(c)) == 0.0


def test_normalize_and_mask_rules(orc):
    rng = np.random.default_rng(9)
    N, M = 400, 260
    raw = rng.uniform(50, 150, (N, M))
    raw[200, 130 + 40] = 1000.0                       # hot pixel -> 12x12 eraser
    norm = orc.normalize(raw)
    mean = raw.mean(); mn = raw.min()
    exp = np.clip(np.rint((raw - mn) / (orc.lib().orc_mean(orc.dp(raw), N, M) * 2.5 - mn) * 255.0), 0, 255)
    assert abs(orc.lib().orc_mean(orc.dp(raw), N, M) - mean) < 1e-9
    assert (norm == exp.astype(np.uint8)).all() and norm[200, 170] == 255
    m = orc.mask(raw)
    assert m[200 - 6:200 + 6, 170 - 6:170 + 6].max() == 0 and m[200 + 6, 170] == 255 and m[200 - 7, 170] == 255
    assert m[:150].max() == 0 and m[N - 149:].max() == 0 and m[150, 100] == 255 and m[N - 150, 100] == 255
    assert m[160, :90].max() == 0 and m[160, 90] == 255 and m[160, M - 90] == 255 and m[160, M - 89] == 0
    assert m[160, 130 - 9:130 + 10].max() == 0 and m[160, 130 - 10] == 255 and m[160, 130 + 10] == 255


# quadtree: quota respected, best response kept per node, deterministic
def test_quadtree_distribution(orc):
    rng = np.random.default_rng(11)
    n = 3000
    pts = set()
    while len(pts) < n:
        pts.add((int(rng.integers(3, 600)), int(rng.integers(3, 900))))
    pts = sorted(pts, key=lambda p: (p[1], p[0]))
    xs = np.array([p[0] for p in pts], np.float32); ys = np.array([p[1] for p in pts], np.float32)
    resp = rng.integers(7, 200, n).astype(np.float32)
    keep = np.zeros(n, np.int32)
    k = orc.lib().orc_quadtree(orc.fp(xs), orc.fp(ys), orc.fp(resp), n, 16, 16 + 603, 16, 16 + 903, 300, orc.ip(keep))
    assert 300 <= k <= 303 + 3 and len(set(keep[:k].tolist())) == k
    keep2 = np.zeros(n, np.int32)
    k2 = orc.lib().orc_quadtree(orc.fp(xs), orc.fp(ys), orc.fp(resp), n, 16, 16 + 603, 16, 16 + 903, 300, orc.ip(keep2))
    assert k2 == k and (keep[:k] == keep2[:k]).all()
    # few points: every point is its own node
    k3 = orc.lib().orc_quadtree(orc.fp(xs[:50]), orc.fp(ys[:50]), orc.fp(resp[:50]), 50, 16, 619, 16, 919, 300, orc.ip(keep))
    assert k3 == 50
    # tall window (the reference's nIni = 0 case) and wide window (nIni = 3)
    k4 = orc.lib().orc_quadtree(orc.fp(xs), orc.fp(ys), orc.fp(resp), n, 0, 300, 0, 1000, 100, orc.ip(keep))
    assert k4 >= 100
    k5 = orc.lib().orc_quadtree(orc.fp(ys), orc.fp(xs), orc.fp(resp), n, 0, 1000, 0, 300 + 300, 100, orc.ip(keep))
    assert k5 >= 100
