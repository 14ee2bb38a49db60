"""Independent witnesses for the oracle's out-of-tree restatements (OpenCV / GTSAM are not in the image, so the oracle
cannot be pinned against them): the same quantities computed by unrelated code that IS in the image -- scipy's matrix
exponential / logarithm and Rotation for the Pose3 algebra, scipy.optimize.least_squares for the LM optima, a brute-force
numpy FAST-9/16 written from the published definition, and a dense numpy Gauss-Newton for the pose graph.  Parity with the
real libraries stays "unpinned"; these shrink the room for a wrong restatement."""
import ctypes as C
import numpy as np
import pytest
import scipy.linalg as sla
from scipy.optimize import least_squares
from scipy.spatial.transform import Rotation


def _hat6(xi):
    w, v = xi[:3], xi[3:]
    X = np.zeros((4, 4))
    X[:3, :3] = [[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]]
    X[:3, 3] = v
    return X


def _T(P):
    M = np.eye(4); M[:3, :3] = np.array(P.R).reshape(3, 3); M[:3, 3] = P.t
    return M


def _pose_of(orc, M):
    P = orc.Pose()
    for i in range(9):
        P.R[i] = float(M[:3, :3].flat[i])
    for i in range(3):
        P.t[i] = float(M[i, 3])
    return P


def test_pose_exp_log_vs_scipy_expm_logm(orc):
    """Pose3::Expmap([w; v]) is the matrix exponential of the 4x4 twist; Logmap its principal logarithm"""
    rng = np.random.default_rng(11)
    for _ in range(100):
        w = rng.standard_normal(3); w *= rng.uniform(1e-3, 3.0) / np.linalg.norm(w)
        xi = np.concatenate([w, rng.uniform(-20, 20, 3)])
        P = orc.Pose(); orc.lib().orc_pose_exp(orc.dp(np.ascontiguousarray(xi)), C.byref(P))
        E = sla.expm(_hat6(xi))
        assert np.abs(_T(P) - E).max() < 1e-11
        assert np.abs(np.array(P.R).reshape(3, 3) - Rotation.from_rotvec(w).as_matrix()).max() < 1e-12
        back = np.zeros(6); orc.lib().orc_pose_log(C.byref(P), orc.dp(back))
        L = np.real(sla.logm(E))
        assert np.abs(back - np.array([L[2, 1], L[0, 2], L[1, 0], L[0, 3], L[1, 3], L[2, 3]])).max() < 1e-8


def test_adjoint_vs_matrix_conjugation(orc):
    """AdjointMap(T) xi == vee(T hat(xi) T^-1), exactly linear (no small-angle argument)"""
    rng = np.random.default_rng(12)
    for _ in range(20):
        xi0 = np.concatenate([rng.standard_normal(3), rng.uniform(-5, 5, 3)])
        P = orc.Pose(); orc.lib().orc_pose_exp(orc.dp(np.ascontiguousarray(xi0)), C.byref(P))
        Ad = np.zeros(36); orc.lib().orc_pose_adjoint(C.byref(P), orc.dp(Ad)); Ad = Ad.reshape(6, 6)
        xi = rng.standard_normal(6)
        Tm = _T(P)
        Cj = Tm @ _hat6(xi) @ np.linalg.inv(Tm)
        assert np.abs(Ad @ xi - np.array([Cj[2, 1], Cj[0, 2], Cj[1, 0], Cj[0, 3], Cj[1, 3], Cj[2, 3]])).max() < 1e-10


def test_compose_between_inverse_vs_matrices(orc):
    rng = np.random.default_rng(13)
    for _ in range(20):
        A = sla.expm(_hat6(rng.standard_normal(6))); B = sla.expm(_hat6(rng.standard_normal(6)))
        PA, PB = _pose_of(orc, A), _pose_of(orc, B)
        o = orc.Pose()
        orc.lib().orc_pose_compose(C.byref(PA), C.byref(PB), C.byref(o)); assert np.abs(_T(o) - A @ B).max() < 1e-12
        orc.lib().orc_pose_between(C.byref(PA), C.byref(PB), C.byref(o)); assert np.abs(_T(o) - np.linalg.inv(A) @ B).max() < 1e-12
        orc.lib().orc_pose_inverse(C.byref(PA), C.byref(o)); assert np.abs(_T(o) - np.linalg.inv(A)).max() < 1e-12


def _tri_residual(p, kp, Ts, Tt, ini, sig_p):
    """LMTriaFactor x 2 + point prior, written from LMtriangulatefactor.cpp:10-27 / optimizer.cpp:984-1021 with numpy only"""
    out = []
    for T, slant in ((Ts, kp[2]), (Tt, kp[5])):
        ps = T[:3, :3].T @ (p - T[:3, 3])
        out += [(np.linalg.norm(ps) - slant) / 0.1, ps[0] / (slant * 0.1 * 3.14159265359 / 180)]
    return np.array(out + list((p - ini) / sig_p))


def test_triangulation_optimum_vs_scipy_least_squares(orc):
    """the GTSAM-style LM of orc_triangulate_one stops within its 1e-5 relative tolerance of the least-squares optimum that
    scipy finds for the same residual vector"""
    rng = np.random.default_rng(14)
    for _ in range(10):
        lm = np.array([rng.uniform(-2, 2), rng.uniform(3, 8), -9.0 + rng.uniform(-0.5, 0.5)])
        Ts = sla.expm(_hat6(np.array([0, 0, rng.uniform(-0.1, 0.1), 0, 0, 0]))); Ts[:3, 3] = [lm[0] + rng.normal(0, 0.01), 0, 0]
        Tt = sla.expm(_hat6(np.array([0, 0, rng.uniform(-0.1, 0.1), 0, 0, 0]))); Tt[:3, 3] = [lm[0] + rng.normal(0, 0.01), 11.0, 0]
        kp = np.array([0, 0, np.linalg.norm(lm - Ts[:3, 3]), 0, 0, np.linalg.norm(lm - Tt[:3, 3]), 0.0])
        ini = lm + rng.normal(0, 0.2, 3)
        sig_p = np.array([10.0, 10.0, np.hypot(*(Ts[:2, 3] - Tt[:2, 3])) / 100])
        got, iters = orc.triangulate_one(kp, np.concatenate([Ts[:3, :3].ravel(), Ts[:3, 3]]), np.concatenate([Tt[:3, :3].ravel(), Tt[:3, 3]]), ini)
        ref = least_squares(_tri_residual, ini, args=(kp, Ts, Tt, ini, sig_p), xtol=1e-14, ftol=1e-14, gtol=1e-14)
        c_got = 0.5 * (_tri_residual(got, kp, Ts, Tt, ini, sig_p) ** 2).sum()
        assert iters >= 1 and c_got <= ref.cost * (1 + 1e-4) + 1e-12
        assert np.abs(got - ref.x).max() < 5e-3


def _fast9_bruteforce(img, thr):
    """corner iff >= 9 contiguous ring pixels are all > v + thr or all < v - thr; score = largest threshold that still
    passes; strict 3x3 NMS, from the definition, no tricks"""
    ring = [(0, 3), (1, 3), (2, 2), (3, 1), (3, 0), (3, -1), (2, -2), (1, -3), (0, -3), (-1, -3), (-2, -2), (-3, -1), (-3, 0), (-3, 1), (-2, 2), (-1, 3)]
    H, W = img.shape
    score = np.zeros((H, W), int)
    I = img.astype(int)
    for y in range(3, H - 3):
        for x in range(3, W - 3):
            d = np.array([I[y + dy, x + dx] - I[y, x] for dx, dy in ring])
            best = 0
            for s in range(16):
                arc = d[(s + np.arange(9)) % 16]
                best = max(best, arc.min(), (-arc).min())          # all brighter by >= arc.min(), or all darker by >= (-arc).min()
            t = best - 1                                           # strict inequality: passes thresholds < best
            if t >= thr:
                score[y, x] = t
    out = []
    for y in range(3, H - 3):
        for x in range(3, W - 3):
            s = score[y, x]
            if s and all(s > score[y + j, x + i] for j in (-1, 0, 1) for i in (-1, 0, 1) if (i or j)):
                out.append((x, y, s))
    return out


def test_fast_cell_vs_bruteforce_numpy(orc):
    """one FAST window of the oracle (cv::FAST(..., thr, true) semantics) against the brute-force definition"""
    rng = np.random.default_rng(15)
    lib = orc.lib()
    for thr in (12, 7):
        img = np.ascontiguousarray((rng.integers(0, 256, (40, 44)) // 4 * 4).astype(np.uint8))
        cap = 4096
        xs = np.zeros(cap, np.int32); ys = np.zeros(cap, np.int32); sc = np.zeros(cap, np.int32)
        n = lib.orc_fast_window(orc.u8(img), img.shape[1], img.shape[0], img.shape[1], thr, orc.ip(xs), orc.ip(ys), orc.ip(sc), cap)
        got = sorted(zip(xs[:n].tolist(), ys[:n].tolist(), sc[:n].tolist()), key=lambda t: (t[1], t[0]))
        ref = sorted(_fast9_bruteforce(img, thr), key=lambda t: (t[1], t[0]))
        assert len(ref) > 5 and got == ref


def test_posegraph_vs_dense_numpy_gauss_newton(orc):
    """the oracle's sparse LM against a dense numpy Gauss-Newton on the same factors (numeric Jacobians of the same
    residual definitions): both reach the same optimum"""
    n = 24
    rng = np.random.default_rng(16)
    dr = np.zeros((n, 6)); dr[:, 3] = 0.05 * np.arange(n); dr[:, 4] = 0.01 * np.sin(np.arange(n)); dr[:, 2] = 0.02 * np.cos(np.arange(n) / 5)
    e = np.zeros(2, orc.LCEDGE_DTYPE)
    for k, (a, b) in enumerate(((2, 17), (5, 22))):
        Ta = sla.expm(_hat6(np.concatenate([dr[a, :3], [0, 0, 0]]))); Ta[:3, 3] = dr[a, 3:]
        Tb = sla.expm(_hat6(np.concatenate([dr[b, :3], [0, 0, 0]]))); Tb[:3, 3] = dr[b, 3:]
        rel = np.linalg.inv(Ta) @ Tb; rel[:3, 3] += rng.normal(0, 0.02, 3)
        e["a"][k] = a; e["b"][k] = b; e["rel"][k] = np.concatenate([rel[:3, :3].ravel(), rel[:3, 3]]); e["var"][k] = [1e-6, 1e-6, 1e-5, 1e-3, 1e-3, 1e-3]
    p = orc.pg_params(); p.add_noise = 0
    out, stats = orc.pg_solve(dr, e, p)
    PI = 3.14159265359
    so = np.array([0.001 * PI / 180, 0.001 * PI / 180, 0.1 * 0.001 * 10 * PI / 180, 0.01, 0.01, 0.001])

    def vee_log(M):
        L = np.real(sla.logm(M)); return np.array([L[2, 1], L[0, 2], L[1, 0], L[0, 3], L[1, 3], L[2, 3]])
    X0 = []
    for i in range(n):
        T = sla.expm(_hat6(np.concatenate([dr[i, :3], [0, 0, 0]]))); T[:3, 3] = dr[i, 3:]; X0.append(T)
    meas = [X0[0]] + [np.linalg.inv(X0[i - 1]) @ X0[i] for i in range(1, n)]
    rels = []
    for k in range(2):
        R = np.eye(4); R[:3, :3] = e["rel"][k][:9].reshape(3, 3); R[:3, 3] = e["rel"][k][9:]; rels.append(R)

    def resid(x):
        X = [X0[i] @ sla.expm(_hat6(x[6 * i:6 * i + 6])) for i in range(n)]
        r = [vee_log(np.linalg.inv(meas[0]) @ X[0]) / 1e-6]
        for i in range(1, n):
            r.append(vee_log(np.linalg.inv(meas[i]) @ np.linalg.inv(X[i - 1]) @ X[i]) / so)
        for k in range(2):
            r.append(vee_log(np.linalg.inv(rels[k]) @ np.linalg.inv(X[e["a"][k]]) @ X[e["b"][k]]) / np.sqrt(e["var"][k]))
        return np.concatenate(r)
    sol = least_squares(resid, np.zeros(6 * n), xtol=1e-15, ftol=1e-15, gtol=1e-15, x_scale=1.0)
    Xs = [X0[i] @ sla.expm(_hat6(sol.x[6 * i:6 * i + 6])) for i in range(n)]
    ref = np.array([np.concatenate([T[:3, :3].ravel(), T[:3, 3]]) for T in Xs])
    assert 0.5 * (resid(np.zeros(6 * n)) ** 2).sum() > 1.5 * sol.cost          # the loop closures really move the optimum off DR
    assert np.abs(out - ref).max() < 2e-5, np.abs(out - ref).max()          # the oracle stops at GTSAM's relative tolerance 1e-5
    assert abs(stats[2] - sol.cost) <= 1e-3 * max(sol.cost, 1e-12) + 1e-9


# ---------------------------------------------------------------------------------------------------------------------
# Round 3: three more witnesses for the OpenCV primitives the oracle restates (none of them is OpenCV: parity stays unpinned,
# but each is an INDEPENDENT derivation -- another language, another structure, or the continuous definition with an error bound).
def _np_resize_linear_u8(src, dh, dw):
    """cv::resize(INTER_LINEAR) for CV_8UC1 written from the published algorithm, vectorised: float32 source
    coordinates at half-pixel centres, 11-bit coefficients rounded half to even, integer horizontal pass, the >>4 / >>16 / +2 >>2
    vertical pass."""
    sh, sw = src.shape
    def table(d, s, clamp):
        scale = 1.0 / (np.float64(d) / np.float64(s))
        f = ((np.arange(d, dtype=np.float64) + 0.5) * scale - 0.5).astype(np.float32)
        i = np.floor(f).astype(np.int64)
        f = (f - i.astype(np.float32)).astype(np.float32)
        if clamp:
            lo = i < 0; f[lo] = 0; i[lo] = 0
            hi = i >= s - 1; f[hi] = 0; i[hi] = s - 1
        c1 = np.rint(f * np.float32(2048)).astype(np.int64); c0 = np.rint((np.float32(1) - f) * np.float32(2048)).astype(np.int64)
        return i, c0, c1
    sx, a0, a1 = table(dw, sw, True)
    sy, b0, b1 = table(dh, sh, False)
    S = src.astype(np.int64)
    H = S[:, sx] * a0[None, :] + S[:, np.minimum(sx + 1, sw - 1)] * a1[None, :]           # sh x dw, scaled by 2^11
    ya = np.clip(sy, 0, sh - 1); yb = np.clip(sy + 1, 0, sh - 1)
    out = (((b0[:, None] * (H[ya] >> 4)) >> 16) + ((b1[:, None] * (H[yb] >> 4)) >> 16) + 2) >> 2
    return out.astype(np.uint8)


def _exact_bilinear(src, dh, dw):
    sh, sw = src.shape
    fy = np.clip((np.arange(dh) + 0.5) * sh / dh - 0.5, 0, sh - 1); fx = np.clip((np.arange(dw) + 0.5) * sw / dw - 0.5, 0, sw - 1)
    y0 = np.floor(fy).astype(int); x0 = np.floor(fx).astype(int); y1 = np.minimum(y0 + 1, sh - 1); x1 = np.minimum(x0 + 1, sw - 1)
    wy = (fy - y0)[:, None]; wx = (fx - x0)[None, :]
    S = src.astype(np.float64)
    return (S[y0][:, x0] * (1 - wx) + S[y0][:, x1] * wx) * (1 - wy) + (S[y1][:, x0] * (1 - wx) + S[y1][:, x1] * wx) * wy


@pytest.mark.parametrize("shape", [(700, 480), (583, 400), (97, 131), (2000, 1024)])
def test_resize_vs_independent_numpy_and_exact_bilinear(orc, shape):
    import ctypes as C
    rng = np.random.default_rng(shape[0])
    src = rng.integers(0, 256, shape, dtype=np.uint8)
    inv = np.float32(1.0) / np.float32(1.2)                                    # mvInvScaleFactor[1] (ORBextractor.cpp:417-424)
    dh, dw = int(np.rint(np.float32(shape[0]) * inv)), int(np.rint(np.float32(shape[1]) * inv))
    dst = np.zeros((dh, dw), np.uint8)
    orc.lib().orc_resize_linear_u8(orc.u8(src), shape[0], shape[1], orc.u8(dst), dh, dw)
    assert (dst == _np_resize_linear_u8(src, dh, dw)).all()                    # bit for bit
    # and the fixed-point result is the continuous bilinear interpolant to within one grey level (11-bit weights, two roundings)
    assert np.abs(dst.astype(np.float64) - _exact_bilinear(src, dh, dw)).max() <= 1.0
    # smooth input: the same bound, and the down-scaled image keeps the mean
    yy, xx = np.mgrid[0:shape[0], 0:shape[1]]
    smooth = (127 + 100 * np.sin(yy / 23.0) * np.cos(xx / 31.0)).astype(np.uint8)
    orc.lib().orc_resize_linear_u8(orc.u8(smooth), shape[0], shape[1], orc.u8(dst), dh, dw)
    assert np.abs(dst.astype(np.float64) - _exact_bilinear(smooth, dh, dw)).max() <= 1.0
    assert abs(dst.mean() - smooth.mean()) < 0.6


def test_fast_atan2_vs_numpy_arctan2(orc):
    """cv::fastAtan2's documented accuracy is 0.3 degrees; the restated polynomial must stay inside it against np.arctan2 on a
    dense set of directions and magnitudes, hit the axes exactly and keep its range [0, 360)"""
    import ctypes as C
    L = orc.lib()
    L.orc_fast_atan2.argtypes = [C.c_float, C.c_float]
    rng = np.random.default_rng(4)
    worst = 0.0
    for k in range(20000):
        ang = rng.uniform(0, 2 * np.pi); r = 10 ** rng.uniform(-3, 6)
        y, x = np.float32(r * np.sin(ang)), np.float32(r * np.cos(ang))
        got = L.orc_fast_atan2(float(y), float(x))
        ref = np.degrees(np.arctan2(float(y), float(x))) % 360.0
        d = abs(got - ref); d = min(d, 360.0 - d)
        worst = max(worst, d)
        assert 0.0 <= got <= 360.0
    assert worst < 0.3, worst
    assert worst > 1e-4                                                       # it IS the polynomial, not libm
    for y, x, ref in ((0, 1, 0.0), (1, 0, 90.0), (0, -1, 180.0), (-1, 0, 270.0), (0, 0, 0.0)):
        assert abs(L.orc_fast_atan2(float(y), float(x)) - ref) < 1e-4
    for y, x in ((1, 1), (1, -1), (-1, -1), (-1, 1)):                         # diagonals: within the polynomial's error of 45 + 90 k
        assert abs(L.orc_fast_atan2(float(y), float(x)) - (np.degrees(np.arctan2(y, x)) % 360)) < 0.3


def test_gaussian_taps_derived_in_the_test(orc):
    """The 8.8 taps of the 13 x 13, sigma 2 blur, derived here and not copied: error diffusion from the edge inwards means the running
    sum of the integer taps is the ROUNDED running sum of the real ones (256 g_i / sum g), so tap_i = rint(C_i) - rint(C_(i-1));
    the centre takes what is left of 256.  Also: symmetric, every tap within one of its own rounding, and the blur it defines is the
    real Gaussian filter to within one grey level."""
    import ctypes as C
    taps = (C.c_int * 13)()
    orc.lib().orc_gauss13_taps(taps)
    taps = np.array(list(taps))
    d = np.arange(13) - 6.0
    g = np.exp(-d * d / (2 * 2.0 ** 2)); g = 256.0 * g / g.sum()
    Cs = np.concatenate([[0.0], np.cumsum(g[:6])])
    assert np.abs(np.abs(Cs[1:] - np.floor(Cs[1:])) - 0.5).min() > 1e-6       # no running sum sits on a rounding tie
    derived = np.diff(np.rint(Cs)).astype(int)
    expect = np.concatenate([derived, [256 - 2 * derived.sum()], derived[::-1]])
    assert (taps == expect).all(), (taps, expect)
    assert taps.sum() == 256 and (taps == taps[::-1]).all() and (np.diff(taps[:7]) > 0).all()
    assert np.abs(taps - g).max() < 1.0                                       # each tap within one unit of the real 8.8 value
    # the filter it defines vs the real-valued separable Gaussian with the same reflect-101 border
    rng = np.random.default_rng(2)
    img = rng.integers(0, 256, (90, 120), dtype=np.uint8)
    out = np.zeros_like(img)
    orc.lib().orc_blur13(orc.u8(img), 90, 120, orc.u8(out))
    gn = g / 256.0
    pad = np.pad(img.astype(np.float64), 6, mode="reflect")
    h = sum(gn[k] * pad[:, k:k + 120] for k in range(13))
    v = sum(gn[k] * h[k:k + 90, :] for k in range(13))
    # error bound from the taps themselves: each pass is off by at most 255 * sum |t_k / 256 - g_k|, plus the final rounding
    eps = np.abs(taps / 256.0 - gn).sum()
    bound = 255.0 * (2 * eps + eps * eps) + 0.5
    err = np.abs(out.astype(np.float64) - v).max()
    assert err <= bound and err <= 3.0, (err, bound)                         # worst case from the taps; white noise stays well inside it
    assert abs(out.mean() - v.mean()) < 0.1                                   # the taps sum to exactly 256: no brightness drift


def test_sincos_polynomial_vs_numpy(orc):
    """The deterministic sin / cos that stands in for libm (frame.cpp:141-149 geo image, ORBextractor.cpp:113 through a float cast):
    Cody-Waite reduction + the fdlibm kernels without FMA.  Against numpy's libm on a dense set of angles -- headings, headings +- pi / 2,
    descriptor angles in radians -- it must stay within an ulp or two of the correctly rounded value, and be exact at 0."""
    L = orc.lib()
    rng = np.random.default_rng(6)
    xs = np.concatenate([rng.uniform(-4 * np.pi, 4 * np.pi, 120000), np.linspace(-np.pi, np.pi, 20001),
                         np.deg2rad(np.arange(0, 360, 0.01)), rng.uniform(-1e-3, 1e-3, 5000)])
    s = C.c_double(0); c = C.c_double(0)
    worst = 0.0
    for x in xs:
        L.orc_sincos(float(x), C.byref(s), C.byref(c))
        worst = max(worst, abs(s.value - np.sin(x)), abs(c.value - np.cos(x)))
    assert worst < 4.5e-16, worst                                             # |sin|, |cos| <= 1: 2 ulp of 1.0
    L.orc_sincos(0.0, C.byref(s), C.byref(c))
    assert s.value == 0.0 and c.value == 1.0


def test_brief_rotation_rounding_vs_float32_libm():
    """computeOrbDescriptor rotates the 512 pattern points by (cosf, sinf) of the keypoint angle and rounds (ORBextractor.cpp:108-126);
    the oracle and the kernels take (float) of the deterministic double sin / cos instead.  How often can that change a descriptor?
    For every angle fastAtan2 can return to 0.01 degree and every pattern point: the rounded coordinates from the two kinds of sin / cos.
    They differ only where a float sin / cos differs in its last bit AND the rotated coordinate sits on a rounding boundary."""
    import os
    pat = np.array([int(t) for t in open(os.path.join(os.path.dirname(__file__), "..", "oracle", "orb_pattern_31.inc")).read().replace("\n", "").split(",") if t.strip()],
                   np.float32).reshape(-1, 2)
    assert pat.shape == (512, 2)
    import ctypes as C2
    from oracle import binding as orc
    L = orc.lib()
    factor = np.float32(np.float32(3.1415926535897932384626433832795) / np.float32(180.0))
    ang = (np.arange(0, 36000, dtype=np.float32) * np.float32(0.01)) * factor          # float angle in radians, as the reference forms it
    s = C2.c_double(0); c = C2.c_double(0)
    a_poly = np.empty(len(ang), np.float32); b_poly = np.empty(len(ang), np.float32)
    for i, x in enumerate(ang):
        L.orc_sincos(float(x), C2.byref(s), C2.byref(c))
        a_poly[i] = np.float32(c.value); b_poly[i] = np.float32(s.value)
    a_libm = np.cos(ang); b_libm = np.sin(ang)                                          # float32 in, float32 out: the float routines
    assert a_libm.dtype == np.float32
    last_bit = np.mean((a_poly != a_libm) | (b_poly != b_libm))
    x0 = pat[:, 0][None, :]; y0 = pat[:, 1][None, :]
    def coords(a, b):
        a = a[:, None]; b = b[:, None]
        return np.rint(x0 * b + y0 * a).astype(np.int32), np.rint(x0 * a - y0 * b).astype(np.int32)     # float32 products and sums, round half to even
    r1, c1 = coords(a_poly, b_poly); r2, c2 = coords(a_libm, b_libm)
    differ = (r1 != r2) | (c1 != c2)
    frac = differ.mean()
    assert last_bit < 0.5                                                     # the float of the double result is the correctly rounded one almost always
    assert frac < 2e-5, (frac, last_bit)                                      # < 0.01 pattern points of a descriptor on average: at most a stray bit
    assert np.abs(r1 - r2).max() <= 1 and np.abs(c1 - c2).max() <= 1
