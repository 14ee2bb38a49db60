"""ctypes binding of the CPU oracle (oracle/liboracle.so).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product package (diasss_amd) never does.
"""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

c_dp = C.POINTER(C.c_double)
c_ip = C.POINTER(C.c_int32)
c_u8p = C.POINTER(C.c_uint8)
c_fp = C.POINTER(C.c_float)


class KP(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("size", C.c_float), ("angle", C.c_float),
                ("response", C.c_float), ("octave", C.c_int32)]


KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"),
                     ("response", "<f4"), ("octave", "<i4")])


class MaskParams(C.Structure):
    _fields_ = [("factor", C.c_double), ("width", C.c_int), ("r", C.c_int), ("side", C.c_int)]


class OrbParams(C.Structure):
    _fields_ = [("nfeatures", C.c_int), ("scale", C.c_float), ("nlevels", C.c_int),
                ("ini_th", C.c_int), ("min_th", C.c_int)]


class MatchParams(C.Structure):
    _fields_ = [("use_l2", C.c_int), ("radius", C.c_double), ("bound_same", C.c_int), ("bound_diff", C.c_int),
                ("l2_bound", C.c_double), ("ratio", C.c_double), ("scc_iters", C.c_int), ("pix_err", C.c_double),
                ("merge_thr", C.c_double)]


class LC(C.Structure):
    _fields_ = [("rel", C.c_double * 12), ("var", C.c_double * 6), ("score", C.c_double), ("iters", C.c_int),
                ("err0", C.c_double), ("err1", C.c_double)]


LC_DTYPE = np.dtype([("rel", "<f8", (12,)), ("var", "<f8", (6,)), ("score", "<f8"), ("iters", "<i4"),
                     ("_pad", "<i4"), ("err0", "<f8"), ("err1", "<f8")])


class LCEdge(C.Structure):
    _fields_ = [("a", C.c_int), ("b", C.c_int), ("rel", C.c_double * 12), ("var", C.c_double * 6)]


LCEDGE_DTYPE = np.dtype([("a", "<i4"), ("b", "<i4"), ("rel", "<f8", (12,)), ("var", "<f8", (6,))])


class PGParams(C.Structure):
    _fields_ = [("max_iters", C.c_int), ("rel_tol", C.c_double), ("abs_tol", C.c_double), ("lambda0", C.c_double),
                ("lambda_factor", C.c_double), ("lambda_max", C.c_double), ("min_fidelity", C.c_double),
                ("add_noise", C.c_int)]


class Pose(C.Structure):
    _fields_ = [("R", C.c_double * 9), ("t", C.c_double * 3)]


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith((".c", ".h", ".inc"))]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        L = _LIB
        L.orc_mean.restype = C.c_double
        L.orc_fast_atan2.restype = C.c_float
        L.orc_fast_atan2.argtypes = [C.c_float, C.c_float]
        L.orc_cvround.argtypes = [C.c_double]
        L.orc_cvroundf.argtypes = [C.c_float]
        L.orc_sincos.argtypes = [C.c_double, c_dp, c_dp]
        L.orc_cvrng_next.restype = C.c_uint32
        L.orc_cvrng_next.argtypes = [C.POINTER(C.c_uint64)]
        L.orc_overlap.restype = C.c_float
        L.orc_ic_angle.restype = C.c_float
    return _LIB


def _p(a, t):
    return a.ctypes.data_as(t)


def dp(a):
    assert a.dtype == np.float64 and a.flags.c_contiguous
    return _p(a, c_dp)


def ip(a):
    assert a.dtype == np.int32 and a.flags.c_contiguous
    return _p(a, c_ip)


def u8(a):
    assert a.dtype == np.uint8 and a.flags.c_contiguous
    return _p(a, c_u8p)


def fp(a):
    assert a.dtype == np.float32 and a.flags.c_contiguous
    return _p(a, c_fp)


def kp_ptr(a):
    assert a.dtype == KP_DTYPE and a.flags.c_contiguous
    return _p(a, C.POINTER(KP))


# ---------------------------------------------------------------- convenience wrappers
def mask_params():
    p = MaskParams(); lib().orc_mask_params_default(C.byref(p)); return p


def orb_params():
    p = OrbParams(); lib().orc_orb_params_default(C.byref(p)); return p


def match_params():
    p = MatchParams(); lib().orc_match_params_default(C.byref(p)); return p


def pg_params():
    p = PGParams(); lib().orc_pg_params_default(C.byref(p)); return p


def normalize(raw):
    raw = np.ascontiguousarray(raw, np.float64)
    out = np.empty(raw.shape, np.uint8)
    lib().orc_normalize(dp(raw), raw.shape[0], raw.shape[1], u8(out))
    return out


def mask(raw, params=None):
    raw = np.ascontiguousarray(raw, np.float64)
    out = np.empty(raw.shape, np.uint8)
    p = params or mask_params()
    lib().orc_mask(dp(raw), raw.shape[0], raw.shape[1], C.byref(p), u8(out))
    return out


def geo_bbox(pose6, gr, M):
    pose6 = np.ascontiguousarray(pose6, np.float64); gr = np.ascontiguousarray(gr, np.float64)
    bb = np.empty(4, np.float64)
    lib().orc_geo_bbox(dp(pose6), dp(gr), pose6.shape[0], M, dp(bb))
    return bb


def geo_img(pose6, gr, M):
    """Frame::GetGeoImg (frame.cpp:126-165): the full N x M pair (x, y)"""
    pose6 = np.ascontiguousarray(pose6, np.float64); gr = np.ascontiguousarray(gr, np.float64)
    N = pose6.shape[0]
    gx = np.empty((N, M), np.float64); gy = np.empty((N, M), np.float64)
    lib().orc_geo_img(dp(pose6), dp(gr), N, M, dp(gx), dp(gy))
    return gx, gy


def geo_at_kps(pose6, gr, M, kps):
    """geo lookup at int(pt.y), int(pt.x) (FEAmatcher.cpp:81-82) -> (n,2) f64"""
    pose6 = np.ascontiguousarray(pose6, np.float64); gr = np.ascontiguousarray(gr, np.float64)
    out = np.empty((len(kps), 2), np.float64)
    x = C.c_double(); y = C.c_double()
    L = lib()
    for i in range(len(kps)):
        L.orc_geo_at(dp(pose6), dp(gr), pose6.shape[0], M, int(kps["y"][i]), int(kps["x"][i]), C.byref(x), C.byref(y))
        out[i, 0] = x.value; out[i, 1] = y.value
    return out


def orb_extract(img, params=None, cap=None):
    img = np.ascontiguousarray(img, np.uint8)
    p = params or orb_params()
    cap = cap or (p.nfeatures + 64)
    kps = np.zeros(cap, KP_DTYPE); desc = np.zeros((cap, 32), np.uint8)
    n = lib().orc_orb_extract(u8(img), img.shape[0], img.shape[1], C.byref(p), kp_ptr(kps), u8(desc), cap)
    return kps[:n].copy(), desc[:n].copy()


def orb_extract_sift(img, params=None, cap=None):
    """the extractor with the descriptor of the SIFT call site next to the ORB one: (kps, desc n x 32, desc128 n x 128 u8)"""
    img = np.ascontiguousarray(img, np.uint8)
    p = params or orb_params()
    cap = cap or (p.nfeatures + 64)
    kps = np.zeros(cap, KP_DTYPE); desc = np.zeros((cap, 32), np.uint8); d128 = np.zeros((cap, 128), np.uint8)
    n = lib().orc_orb_extract_sift(u8(img), img.shape[0], img.shape[1], C.byref(p), kp_ptr(kps), u8(desc), u8(d128), cap)
    return kps[:n].copy(), desc[:n].copy(), d128[:n].copy()


def blur13(img):
    img = np.ascontiguousarray(img, np.uint8); out = np.empty_like(img)
    lib().orc_blur13(u8(img), img.shape[0], img.shape[1], u8(out))
    return out


def sift_hist(blurred, x, y, angle_deg):
    """the 4 x 4 x 8 histogram of oracle/orc_sift.c in 2^-12 fixed point (int32[128]) at integer level coordinates"""
    b = np.ascontiguousarray(blurred, np.uint8); h = np.zeros(128, np.int32)
    lib().orc_sift_hist(u8(b), b.shape[0], b.shape[1], int(x), int(y), C.c_float(angle_deg), ip(h))
    return h


def sift_finalize(h):
    h = np.ascontiguousarray(h, np.int32); out = np.zeros(128, np.uint8)
    lib().orc_sift_finalize(ip(h), u8(out))
    return out


def sift128(blurred, x, y, angle_deg):
    return sift_finalize(sift_hist(blurred, x, y, angle_deg))


def sift_weights(n):
    w = np.zeros(n, np.float32)
    lib().orc_sift_weights(fp(w), n)
    return w


def detect_feature(raw, mparams=None, oparams=None, sift=False):
    """Frame::DetectFeature on the normalised image + mask filter (frame.cpp:167-203); sift=True also returns the n x 128 rows"""
    norm = normalize(raw); msk = mask(raw, mparams)
    if sift:
        kps, desc, d128 = orb_extract_sift(norm, oparams)
        n = lib().orc_mask_filter2(kp_ptr(kps), u8(desc), u8(d128), len(kps), u8(msk), raw.shape[1]) if len(kps) else 0
        return kps[:n].copy(), desc[:n].copy(), norm, msk, d128[:n].copy()
    kps, desc = orb_extract(norm, oparams)
    n = lib().orc_mask_filter(kp_ptr(kps), u8(desc), len(kps), u8(msk), raw.shape[1]) if len(kps) else 0
    return kps[:n].copy(), desc[:n].copy(), norm, msk


def robust_matching(id_s, id_t, rows_s, rows_t, kps_s, desc_s, geo_s, bbox_s, kps_t, desc_t, geo_t, bbox_t, params=None):
    p = params or match_params()
    cap = len(kps_s) + len(kps_t) + 1
    rows = np.zeros((cap, 6), np.float64)
    kps_s = np.ascontiguousarray(kps_s); kps_t = np.ascontiguousarray(kps_t)
    dlen = 128 if p.use_l2 == 2 else 32            # use_l2 = 2: rows of the 128-element SIFT descriptor
    desc_s = np.ascontiguousarray(desc_s, np.uint8).reshape(-1, dlen); desc_t = np.ascontiguousarray(desc_t, np.uint8).reshape(-1, dlen)
    assert len(desc_s) == len(kps_s) and len(desc_t) == len(kps_t)
    geo_s = np.ascontiguousarray(geo_s, np.float64).reshape(-1, 2); geo_t = np.ascontiguousarray(geo_t, np.float64).reshape(-1, 2)
    n = lib().orc_robust_matching(id_s, id_t, rows_s, rows_t,
                                  kp_ptr(kps_s), u8(desc_s), dp(geo_s), len(kps_s), dp(np.ascontiguousarray(bbox_s, np.float64)),
                                  kp_ptr(kps_t), u8(desc_t), dp(geo_t), len(kps_t), dp(np.ascontiguousarray(bbox_t, np.float64)),
                                  C.byref(p), dp(rows), cap)
    return rows[:n].copy()


def match_dir(id_a, id_b, rows_b, kps_a, desc_a, geo_a, kps_b, desc_b, geo_b, bbox_b, params=None, scc=True):
    p = params or match_params()
    na = len(kps_a)
    corres = np.full(max(na, 1), -1, np.int32)
    best = np.zeros(max(na, 1), np.int32); second = np.zeros(max(na, 1), np.int32); ncand = np.zeros(max(na, 1), np.int32)
    kps_a = np.ascontiguousarray(kps_a); kps_b = np.ascontiguousarray(kps_b)
    dlen = 128 if p.use_l2 == 2 else 32
    desc_a = np.ascontiguousarray(desc_a, np.uint8).reshape(-1, dlen); desc_b = np.ascontiguousarray(desc_b, np.uint8).reshape(-1, dlen)
    assert len(desc_a) == na and len(desc_b) == len(kps_b)
    geo_a = np.ascontiguousarray(geo_a, np.float64).reshape(-1, 2); geo_b = np.ascontiguousarray(geo_b, np.float64).reshape(-1, 2)
    bb = np.ascontiguousarray(bbox_b, np.float64)
    L = lib()
    L.orc_match_nn(id_a, id_b, kp_ptr(kps_a), u8(desc_a), dp(geo_a), na, kp_ptr(kps_b), u8(desc_b), dp(geo_b), len(kps_b),
                   dp(bb), C.byref(p), ip(corres), ip(best), ip(second), ip(ncand))
    nn = corres[:na].copy()
    cnt = C.c_int(0); model = C.c_double(0.0); hist = 0
    if scc:
        hist = L.orc_match_scc(id_a, id_b, rows_b, kp_ptr(kps_a), na, kp_ptr(kps_b), C.byref(p), ip(corres), C.byref(cnt), C.byref(model))
    return dict(nn=nn, corres=corres[:na].copy(), best=best[:na], second=second[:na], ncand=ncand[:na],
                hist=hist, scc_count=cnt.value, scc_model=model.value)


def get_kps_pairs(rows6, id_t, alt_s, gr_s, alt_t, gr_t):
    rows6 = np.ascontiguousarray(rows6, np.float64).reshape(-1, 6)
    out = np.zeros((max(len(rows6), 1), 7), np.float64)
    alt_s = np.ascontiguousarray(alt_s, np.float64); gr_s = np.ascontiguousarray(gr_s, np.float64)
    alt_t = np.ascontiguousarray(alt_t, np.float64); gr_t = np.ascontiguousarray(gr_t, np.float64)
    n = lib().orc_get_kps_pairs(dp(rows6), len(rows6), id_t, dp(alt_s), dp(gr_s), len(gr_s), dp(alt_t), dp(gr_t), len(gr_t),
                                dp(out), len(out))
    return out[:n].copy()


def lc_solve(kp7, pose_s, alt_s, gr_s, Ms, pose_t, alt_t, gr_t, Mt):
    kp7 = np.ascontiguousarray(kp7, np.float64).reshape(-1, 7)
    n = len(kp7)
    out = np.zeros(max(n, 1), LC_DTYPE)
    assert LC_DTYPE.itemsize == C.sizeof(LC)
    pose_s = np.ascontiguousarray(pose_s, np.float64); pose_t = np.ascontiguousarray(pose_t, np.float64)
    alt_s = np.ascontiguousarray(alt_s, np.float64); gr_s = np.ascontiguousarray(gr_s, np.float64)
    alt_t = np.ascontiguousarray(alt_t, np.float64); gr_t = np.ascontiguousarray(gr_t, np.float64)
    lib().orc_lc_solve(dp(kp7), n, dp(pose_s), dp(alt_s), dp(gr_s), pose_s.shape[0], Ms,
                       dp(pose_t), dp(alt_t), dp(gr_t), pose_t.shape[0], Mt, out.ctypes.data_as(C.POINTER(LC)))
    return out[:n].copy()


def triangulate(kp7, pose_s, alt_s, gr_s, Ms, pose_t, alt_t, gr_t, Mt):
    """orc_triangulate: n x 7 = [x y z | range_s plane_s range_t plane_t] (optimizer.cpp:907-921, 984-1021)"""
    kp7 = np.ascontiguousarray(kp7, np.float64).reshape(-1, 7)
    n = len(kp7)
    out = np.zeros((max(n, 1), 7))
    pose_s = np.ascontiguousarray(pose_s, np.float64); pose_t = np.ascontiguousarray(pose_t, np.float64)
    alt_s = np.ascontiguousarray(alt_s, np.float64); gr_s = np.ascontiguousarray(gr_s, np.float64)
    alt_t = np.ascontiguousarray(alt_t, np.float64); gr_t = np.ascontiguousarray(gr_t, np.float64)
    lib().orc_triangulate(dp(kp7), n, dp(pose_s), dp(alt_s), dp(gr_s), pose_s.shape[0], Ms,
                          dp(pose_t), dp(alt_t), dp(gr_t), pose_t.shape[0], Mt, dp(out))
    return out[:n].copy()


def triangulate_one(kp7, Tp_s, Tp_t, lm_ini):
    """orc_triangulate_one with identity sensor offsets; Tp_* = 12 doubles (R row-major, t)"""
    Ts = Pose(); Ts.R[0] = Ts.R[4] = Ts.R[8] = 1.0
    Ps = Pose(); Pt = Pose()
    for k in range(9):
        Ps.R[k] = float(Tp_s[k]); Pt.R[k] = float(Tp_t[k])
    for k in range(3):
        Ps.t[k] = float(Tp_s[9 + k]); Pt.t[k] = float(Tp_t[9 + k])
    kp7 = np.ascontiguousarray(kp7, np.float64); ini = np.ascontiguousarray(lm_ini, np.float64); out = np.zeros(3)
    it = lib().orc_triangulate_one(dp(kp7), C.byref(Ts), C.byref(Ts), C.byref(Ps), C.byref(Pt), dp(ini), dp(out))
    return out, int(it)


def pg_select_lc(frame_rows, pair_s, pair_t, pair_off, kp7, lcs):
    frame_rows = np.ascontiguousarray(frame_rows, np.int32)
    pair_s = np.ascontiguousarray(pair_s, np.int32); pair_t = np.ascontiguousarray(pair_t, np.int32)
    pair_off = np.ascontiguousarray(pair_off, np.int32)
    kp7 = np.ascontiguousarray(kp7, np.float64).reshape(-1, 7)
    lcs = np.ascontiguousarray(lcs)
    cap = max(len(kp7), 1)
    edges = np.zeros(cap, LCEDGE_DTYPE)
    assert LCEDGE_DTYPE.itemsize == C.sizeof(LCEdge)
    n = lib().orc_pg_select_lc(len(frame_rows), ip(frame_rows), len(pair_s), ip(pair_s), ip(pair_t), ip(pair_off),
                               dp(kp7), lcs.ctypes.data_as(C.POINTER(LC)), edges.ctypes.data_as(C.POINTER(LCEdge)), cap)
    return edges[:n].copy()


_REDUCED_FN = C.CFUNCTYPE(C.c_int, C.c_int, C.c_int, c_ip, c_ip, c_dp, c_dp)


def _make_reduced_solver(refine, log):
    """The reduced (separator) system of orc_pg_solve through scipy's sparse LU (SuperLU, symmetric mode) instead of the
    envelope Cholesky: what lets the oracle's own LM run at the size of BASELINE config 3 / 4 in a minute.  `refine`
    steps of iterative refinement with the residual accumulated in long double."""
    import scipy.sparse as sp
    import scipy.sparse.linalg as spla

    def solve(ns, nblk, bi, bj, blk, rhs):
        try:
            bi = np.ctypeslib.as_array(bi, (nblk,)).astype(np.int64)
            bj = np.ctypeslib.as_array(bj, (nblk,)).astype(np.int64)
            B = np.ctypeslib.as_array(blk, (nblk, 6, 6))
            b = np.ctypeslib.as_array(rhs, (ns * 6,))
            a6 = np.arange(6)
            rows = (bi[:, None, None] * 6 + a6[None, :, None]) + np.zeros((1, 1, 6), np.int64)
            cols = (bj[:, None, None] * 6 + a6[None, None, :]) + np.zeros((1, 6, 1), np.int64)
            off = bi != bj
            r = np.concatenate([rows.ravel(), cols[off].ravel()])
            c = np.concatenate([cols.ravel(), rows[off].ravel()])
            v = np.concatenate([B.ravel(), B[off].ravel()])
            A = sp.coo_matrix((v, (r, c)), shape=(ns * 6, ns * 6)).tocsc()
            lu = spla.splu(A, permc_spec="MMD_AT_PLUS_A", diag_pivot_thresh=0.0, options=dict(SymmetricMode=True))
            b0 = b.copy()
            x = lu.solve(b0)
            for _ in range(refine):
                Ac = A.tocoo()
                res = b0.astype(np.longdouble)
                np.subtract.at(res, Ac.row, Ac.data.astype(np.longdouble) * x[Ac.col].astype(np.longdouble))
                x = x + lu.solve(res.astype(np.float64))
            if not np.all(np.isfinite(x)):
                return -1
            b[:] = x
            if log is not None:
                log.append((ns, int(A.nnz), int(lu.L.nnz)))
            return 0
        except Exception as e:          # a Python exception must not unwind through the C frame
            print("oracle reduced solver failed:", repr(e))
            return -1
    return _REDUCED_FN(solve)


def pg_solve(dr, edges, params=None, solver="envelope", refine=0, log=None, full_refine=0):
    """orc_pg_solve.  solver = "envelope" (the C file's own exact skyline Cholesky; small graphs) or "sparse" (the same LM loop,
    chain condensation and back-substitution with the reduced system solved by scipy's sparse LU: full-size graphs)."""
    dr = np.ascontiguousarray(dr, np.float64).reshape(-1, 6)
    p = params or pg_params()
    out = np.zeros((len(dr), 12), np.float64); stats = np.zeros(4, np.float64)
    edges = np.ascontiguousarray(edges)
    L = lib()
    L.orc_pg_set_reduced_solver.argtypes = [C.c_void_p]
    cb = None
    if solver == "sparse":
        cb = _make_reduced_solver(refine, log)
        L.orc_pg_set_reduced_solver(C.cast(cb, C.c_void_p))
    else:
        assert solver == "envelope"
    L.orc_pg_set_full_refine(int(full_refine))
    try:
        L.orc_pg_solve(dp(dr), len(dr), edges.ctypes.data_as(C.POINTER(LCEdge)), len(edges), C.byref(p), dp(out), dp(stats))
    finally:
        L.orc_pg_set_reduced_solver(None)
        L.orc_pg_set_full_refine(0)
    return out, stats


def pg_error_at(dr, edges, poses12):
    """0.5 sum |r|^2 of the pose graph (dr, edges) at the poses `poses12` (total x 12)"""
    dr = np.ascontiguousarray(dr, np.float64).reshape(-1, 6)
    x = np.ascontiguousarray(poses12, np.float64).reshape(-1, 12)
    assert len(x) == len(dr)
    edges = np.ascontiguousarray(edges)
    f = lib().orc_pg_error_at
    f.restype = C.c_double
    return float(f(dp(dr), len(dr), edges.ctypes.data_as(C.POINTER(LCEdge)), len(edges), dp(x)))
