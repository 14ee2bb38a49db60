"""Known-answer tests for the GTSAM-semantics restatement: Pose3 exp/log,
SssPointFactor Jacobians vs central differences (and the documented F6 deviation), mini-LM recovery."""
import ctypes as C
import math
import numpy as np


def _pose(orc, xi):
    T = orc.Pose(); v = np.ascontiguousarray(xi, np.float64)
    orc.lib().orc_pose_exp(orc.dp(v), C.byref(T)); return T


def _log(orc, T):
    o = np.zeros(6); orc.lib().orc_pose_log(C.byref(T), orc.dp(o)); return o


def _retract(orc, T, xi):
    o = orc.Pose(); v = np.ascontiguousarray(xi, np.float64)
    orc.lib().orc_pose_retract(C.byref(T), orc.dp(v), C.byref(o)); return o


def test_pose_log_exp_roundtrip(orc):
    rng = np.random.default_rng(0)
    for _ in range(200):
        w = rng.standard_normal(3); w *= rng.uniform(0, 3.0) / np.linalg.norm(w)       # |omega| < pi
        xi = np.concatenate([w, rng.uniform(-50, 50, 3)])
        assert np.allclose(_log(orc, _pose(orc, xi)), xi, atol=1e-9)
    for xi in ([0, 0, 0, 1, 2, 3], [1e-9, 0, 0, 1, 2, 3], [0, 0, 3.14159, 1, 2, 3], [0, 0, -3.1415926, 5, 0, 0]):
        assert np.allclose(_log(orc, _pose(orc, xi)), xi, atol=1e-6)


def test_rodrigues_is_rotation_vector_and_rpy(orc):
    T = orc.Pose(); p6 = np.array([0.0, 0.0, 0.7, 1.0, 2.0, 3.0])
    orc.lib().orc_pose_from_rodrigues(orc.dp(p6), C.byref(T))
    R = np.array(T.R).reshape(3, 3)
    c, s = math.cos(0.7), math.sin(0.7)
    assert np.allclose(R, [[c, -s, 0], [s, c, 0], [0, 0, 1]], atol=1e-15)
    rpy = np.zeros(3); orc.lib().orc_pose_rpy(C.byref(T), orc.dp(rpy))
    assert np.allclose(rpy, [0, 0, 0.7], atol=1e-15)
    # general rotation: R = Rz(y) Ry(p) Rx(r)
    r_, p_, y_ = 0.3, -0.4, 2.0
    Rx = np.array([[1, 0, 0], [0, math.cos(r_), -math.sin(r_)], [0, math.sin(r_), math.cos(r_)]])
    Ry = np.array([[math.cos(p_), 0, math.sin(p_)], [0, 1, 0], [-math.sin(p_), 0, math.cos(p_)]])
    Rz = np.array([[math.cos(y_), -math.sin(y_), 0], [math.sin(y_), math.cos(y_), 0], [0, 0, 1]])
    Rm = Rz @ Ry @ Rx
    for i in range(9): T.R[i] = Rm.flat[i]
    orc.lib().orc_pose_rpy(C.byref(T), orc.dp(rpy))
    assert np.allclose(rpy, [r_, p_, y_], atol=1e-12)


def test_adjoint_identity(orc):
    """Ad(T) xi == Log(T Exp(xi) T^-1) to first order"""
    rng = np.random.default_rng(1)
    T = _pose(orc, [0.3, -0.2, 0.9, 4, -2, 1])
    Ad = np.zeros(36); orc.lib().orc_pose_adjoint(C.byref(T), orc.dp(Ad)); Ad = Ad.reshape(6, 6)
    Ti = orc.Pose(); orc.lib().orc_pose_inverse(C.byref(T), C.byref(Ti))
    for _ in range(5):
        xi = rng.standard_normal(6) * 1e-6
        A = orc.Pose(); B = orc.Pose()
        E = _pose(orc, xi)
        orc.lib().orc_pose_compose(C.byref(T), C.byref(E), C.byref(A))
        orc.lib().orc_pose_compose(C.byref(A), C.byref(Ti), C.byref(B))
        assert np.allclose(_log(orc, B), Ad @ xi, atol=1e-11)


def _sss(orc, p, T, Ts, mx, my):
    e = np.zeros(2); H1 = np.zeros(6); H2 = np.zeros(12)
    pp = np.ascontiguousarray(p, np.float64)
    orc.lib().orc_sss_factor(orc.dp(pp), C.byref(T), C.byref(Ts), C.c_double(mx), C.c_double(my), orc.dp(e), orc.dp(H1), orc.dp(H2))
    return e, H1.reshape(2, 3), H2.reshape(2, 6)


# (8) SssPointFactor Jacobians (SSSpointfactor.cpp:11-80)
def test_sss_factor_jacobians(orc):
    rng = np.random.default_rng(2)
    T = _pose(orc, [0.05, -0.03, 0.8, 10, -4, 2])
    Ts = _pose(orc, [0, 0, 0, 0, 0, 0])
    p = np.array([14.0, 3.0, -8.0])
    e, H1, H2 = _sss(orc, p, T, Ts, 12.0, 0.0)
    h = 1e-6
    num1 = np.zeros((2, 3))
    for k in range(3):
        d = np.zeros(3); d[k] = h
        num1[:, k] = (_sss(orc, p + d, T, Ts, 12.0, 0.0)[0] - _sss(orc, p - d, T, Ts, 12.0, 0.0)[0]) / (2 * h)
    assert np.allclose(H1, num1, atol=1e-7)
    num2 = np.zeros((2, 6))
    for k in range(6):
        d = np.zeros(6); d[k] = h
        num2[:, k] = (_sss(orc, p, _retract(orc, T, d), Ts, 12.0, 0.0)[0] - _sss(orc, p, _retract(orc, T, -d), Ts, 12.0, 0.0)[0]) / (2 * h)
    assert np.allclose(H2[:, :3], num2[:, :3], atol=1e-6)          # rotation block agrees with the body-frame chart
    # translation block is the reference's -(Rs^T R^T) (world-frame increment), NOT the chart-consistent -Rs^T (F6)
    R = np.array(T.R).reshape(3, 3)
    pm = R.T @ (p - np.array(T.t)); ps = pm
    Jp_t = -(R.T)
    exp_t = np.vstack([ps @ Jp_t / np.linalg.norm(ps), Jp_t[0]])
    assert np.allclose(H2[:, 3:], exp_t, atol=1e-12)
    assert not np.allclose(H2[:, 3:], num2[:, 3:], atol=1e-3)
    # with R = I the two coincide (why the yaw~pi hack keeps the mini problems convergent)
    T0 = _pose(orc, [0, 0, 0, 10, -4, 2])
    _, _, H20 = _sss(orc, p, T0, Ts, 12.0, 0.0)
    for k in range(3, 6):
        d = np.zeros(6); d[k] = h
        col = (_sss(orc, p, _retract(orc, T0, d), Ts, 12.0, 0.0)[0] - _sss(orc, p, _retract(orc, T0, -d), Ts, 12.0, 0.0)[0]) / (2 * h)
        assert np.allclose(H20[:, k], col, atol=1e-7)


# (10) mini-LM on noise-free geometry: the DR poses are exact, so the optimum is the DR relative pose
def test_mini_lm_noise_free(orc):
    N, M = 64, 400; half = M // 2
    gr = 0.05 * np.arange(half)
    alt = np.full(N, 9.0)
    pose_s = np.zeros((N, 6)); pose_s[:, 3] = 0.05 * np.arange(N)                      # heading 0 along +x, y = 0
    pose_t = np.zeros((N, 6)); pose_t[:, 3] = 0.05 * np.arange(N) + 0.4; pose_t[:, 4] = 6.0
    # one seafloor point seen from both tracks: world (1.0, 4.0, -9): ping where x matches, starboard (y+) for s, port for t
    ping_s = 20; bin_s = half + int(round(4.0 / 0.05))          # s at x=1.0, point at +4 m
    ping_t = 12; bin_t = half - int(round(2.0 / 0.05))          # t at x=1.0, y=6: point at -2 m
    rows = np.array([[0, 2, ping_s, bin_s, ping_t, bin_t]], float)
    kp7 = orc.get_kps_pairs(rows, 2, alt, gr, alt, gr)
    assert len(kp7) == 1
    lcs = orc.lc_solve(kp7, pose_s, alt, gr, M, pose_t, alt, gr, M)
    rel = lcs["rel"][0]
    assert np.allclose(rel[:9].reshape(3, 3), np.eye(3), atol=1e-9)
    assert np.allclose(rel[9:], [0.0, 6.0, 0.0], atol=1e-9)     # x: 0.05*12+0.4 - 0.05*20 = 0
    assert lcs["err1"][0] < 1e-18 and np.all(lcs["var"][0] > 0)


def test_mini_lm_pulls_drifted_target(orc):
    """target DR is displaced by 0.5 m ALONG-track (the direction the plane constraint p_s.x = 0 observes; an
    across-track shift is absorbed by the free landmark): the optimised X2 moves back (score > 0)"""
    N, M = 64, 400; half = M // 2
    gr = 0.05 * np.arange(half); alt = np.full(N, 9.0)
    pose_s = np.zeros((N, 6)); pose_s[:, 3] = 0.05 * np.arange(N)
    pose_t = np.zeros((N, 6)); pose_t[:, 3] = 0.05 * np.arange(N) + 0.4 + 0.5; pose_t[:, 4] = 6.0   # drifted DR
    ping_s, bin_s, ping_t, bin_t = 20, half + 80, 12, half - 40     # true geometry as in the noise-free case
    kp7 = orc.get_kps_pairs(np.array([[0, 2, ping_s, bin_s, ping_t, bin_t]], float), 2, alt, gr, alt, gr)
    lcs = orc.lc_solve(kp7, pose_s, alt, gr, M, pose_t, alt, gr, M)
    assert lcs["score"][0] > 0 and lcs["err1"][0] < lcs["err0"][0]
    assert abs(lcs["rel"][0][9]) < 0.1                             # x pulled from 0.5 toward the true 0.0


def test_posegraph_trivial_and_lc(orc):
    n = 40
    dr = np.zeros((n, 6)); dr[:, 3] = 0.05 * np.arange(n)
    p = orc.pg_params(); p.add_noise = 0
    out, stats = orc.pg_solve(dr, np.zeros(0, orc.LCEDGE_DTYPE), p)
    assert np.allclose(out[:, 9], dr[:, 3], atol=1e-12) and stats[2] < 1e-20
    # with the reference's noisy initial values the chain still relaxes back onto DR (no LC => DR is the optimum)
    p.add_noise = 1
    out, stats = orc.pg_solve(dr, np.zeros(0, orc.LCEDGE_DTYPE), p)
    assert np.allclose(out[:, 9:12], dr[:, 3:6], atol=1e-6) and stats[0] >= 1
    # one loop closure that contradicts odometry by 1 cm in y between pose 5 and 30: solution splits the disagreement
    e = np.zeros(1, orc.LCEDGE_DTYPE)
    e["a"] = 5; e["b"] = 30
    rel = np.concatenate([np.eye(3).ravel(), [0.05 * 25, 0.01, 0.0]]); e["rel"][0] = rel
    e["var"][0] = [1e-6, 1e-6, 1e-6, 1e-4, 1e-4, 1e-4]
    out, stats = orc.pg_solve(dr, e, p)
    dy = out[30, 10] - out[5, 10]
    assert 0.0 < dy < 0.01 and abs(out[0, 10]) < 1e-9
