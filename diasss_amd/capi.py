"""ctypes binding of libdsss.so (the C ABI declared in include/dsss.h).

This is the only way Python reaches the hot path: there is no Python or CPU implementation behind it.
If the HIP library has not been built, or no MI355X is visible, loading/creating fails loudly.
"""
import ctypes as C
import os
import subprocess
import sys
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DSSS_LIB", os.path.join(_HERE, "libdsss.so"))     # DSSS_LIB: A/B runs of two builds on one box
_LIB = None
_HIP = None

KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"),
                     ("response", "<f4"), ("octave", "<i4")])
LC_DTYPE = np.dtype([("rel", "<f8", (12,)), ("var", "<f8", (6,)), ("score", "<f8"), ("iters", "<i4"),
                     ("_pad", "<i4"), ("err0", "<f8"), ("err1", "<f8")])
LCEDGE_DTYPE = np.dtype([("a", "<i4"), ("b", "<i4"), ("rel", "<f8", (12,)), ("var", "<f8", (6,))])

K_NAMES = ["row_reduce", "pre_misc", "normalize", "pyramid", "fast", "fast_compact", "desc", "filter", "match", "scc", "rows", "lc", "pg",
           "quadtree", "pg_acc", "pg_diag", "pg_trsm", "pg_bwd", "pg_subtree", "pg_asm", "pg_comm", "pg_rsu", "match_done", "sift"]


class MaskParams(C.Structure):
    _fields_ = [("factor", C.c_double), ("width", C.c_int32), ("r", C.c_int32), ("side", C.c_int32)]


class OrbParams(C.Structure):
    _fields_ = [("nfeatures", C.c_int32), ("scale", C.c_float), ("nlevels", C.c_int32),
                ("ini_th", C.c_int32), ("min_th", C.c_int32), ("descriptor", C.c_int32)]


DESC_ORB, DESC_SIFT128 = 0, 1          # include/dsss.h DSSS_DESC_*


class MatchParams(C.Structure):
    _fields_ = [("use_l2", C.c_int32), ("radius", C.c_double), ("bound_same", C.c_int32), ("bound_diff", C.c_int32),
                ("l2_bound", C.c_double), ("ratio", C.c_double), ("scc_iters", C.c_int32), ("pix_err", C.c_double),
                ("merge_thr", C.c_double)]


class PGParams(C.Structure):
    _fields_ = [("max_iters", C.c_int32), ("rel_tol", C.c_double), ("abs_tol", C.c_double), ("lambda0", C.c_double),
                ("lambda_factor", C.c_double), ("lambda_max", C.c_double), ("min_fidelity", C.c_double),
                ("add_noise", C.c_int32)]


class DsssError(RuntimeError):
    pass


class FramesArgs:
    """argument arrays of dsss_frames_set (see Context.frames_args)"""

    def __init__(self, ids, raws, Ns, Ms, poses, alts, grs):
        n = self.n = len(ids)
        if not (len(raws) == len(Ns) == len(Ms) == len(poses) == len(alts) == len(grs) == n):
            raise ValueError("frames_set: the per-frame lists differ in length")
        _torch = sys.modules.get("torch")                      # (only a caller that has torch can hand tensors over)

        def addr(a, what, shape=None):
            if a is None:
                return 0
            if isinstance(a, np.ndarray):
                if a.dtype != np.float64 or not a.flags.c_contiguous:
                    raise TypeError("frames_set: %s must be a C-contiguous float64 array" % what)
                if shape is not None and a.shape != shape:
                    raise ValueError("frames_set: %s has shape %s, expected %s" % (what, a.shape, shape))
                return a.__array_interface__["data"][0]
            if _torch is not None and isinstance(a, _torch.Tensor):
                if a.dtype != _torch.float64 or not a.is_contiguous():
                    raise TypeError("frames_set: %s must be a contiguous float64 tensor" % what)
                if shape is not None and tuple(a.shape) != shape:
                    raise ValueError("frames_set: %s has shape %s, expected %s" % (what, tuple(a.shape), shape))
                return a.data_ptr()                            # host (pinned or pageable) or device: the library asks the runtime which
            raise TypeError("frames_set: %s must be None, a numpy array or a torch tensor" % what)
        self.ids = np.ascontiguousarray(ids, np.int32); self.N = np.ascontiguousarray(Ns, np.int32); self.M = np.ascontiguousarray(Ms, np.int32)
        self.p_raw = np.fromiter((addr(a, "raw[%d]" % i, (int(self.N[i]), int(self.M[i]))) for i, a in enumerate(raws)), np.uintp, n)
        self.p_pose = np.fromiter((addr(a, "pose[%d]" % i, (int(self.N[i]), 6)) for i, a in enumerate(poses)), np.uintp, n)
        self.p_alt = np.fromiter((addr(a, "alt[%d]" % i, (int(self.N[i]),)) for i, a in enumerate(alts)), np.uintp, n)
        self.p_gr = np.fromiter((addr(a, "grange[%d]" % i, (int(self.M[i]) // 2,)) for i, a in enumerate(grs)), np.uintp, n)
        if (self.p_pose == 0).any() or (self.p_alt == 0).any() or (self.p_gr == 0).any():
            raise ValueError("frames_set: pose, altitude and ground range are required for every frame")
        self.keep = dict(zip(map(int, ids), zip(raws, poses, alts, grs)))      # the arrays stay alive as long as this object (or the Context) does


def build(force=False):
    """compile libdsss.so for gfx950 with hipcc (cross-compiles without a GPU)"""
    src = os.path.join(_HERE, "csrc")
    if force:
        subprocess.check_call(["make", "-C", src, "-s", "clean"])
    subprocess.check_call(["make", "-C", src, "-s", "-j4"])
    return LIB_PATH


def _load_hip_runtime():
    """libdsss.so leaves the HIP runtime symbols undefined so that ONE runtime serves the whole process.
    PyTorch ships its own libamdhip64 (soname libamdhip64.so) next to the system one (libamdhip64.so.7); loading
    both makes the second one see no GPU.  So: if torch is importable use the copy it loads, else the system copy."""
    try:
        import torch  # noqa: F401  (loads torch/lib/libamdhip64.so)
        cand = os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so")
        if os.path.exists(cand):
            return C.CDLL(cand, mode=C.RTLD_GLOBAL)
    except Exception:
        pass
    for cand in ("/opt/rocm/lib/libamdhip64.so", "libamdhip64.so.7", "libamdhip64.so"):
        try:
            return C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            continue
    raise DsssError("no HIP runtime (libamdhip64) found")


def lib():
    global _LIB, _HIP
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise DsssError("libdsss.so is not built (run `python -c 'import __graft_entry__ as g; g.build()'`); "
                            "there is no CPU fallback")
        _HIP = _load_hip_runtime()
        L = C.CDLL(LIB_PATH)
        L.dsss_strerror.restype = C.c_char_p
        L.dsss_last_error.restype = C.c_char_p
        L.dsss_last_error.argtypes = [C.c_void_p]
        L.dsss_stream.restype = C.c_void_p
        L.dsss_stream.argtypes = [C.c_void_p]
        L.dsss_features_pack_bytes.restype = C.c_size_t
        L.dsss_features_pack_bytes.argtypes = [C.c_void_p]
        L.dsss_comm_init_callback.argtypes = [C.c_void_p, C.c_int, C.c_int, COMM_FN, C.c_void_p]
        L.dsss_comm_init_device_callback.argtypes = [C.c_void_p, C.c_int, C.c_int, COMM_DEV_FN, C.c_void_p]
        L.dsss_comm_frame_owner.argtypes = [C.c_void_p, C.c_int, C.c_int]
        _LIB = L
    return _LIB


COMM_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_size_t)
COMM_DEV_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p)


def _ptr(a):
    """host numpy array, torch tensor (host or device) or raw int address -> void*"""
    if a is None:
        return None
    if isinstance(a, np.ndarray):
        assert a.flags.c_contiguous
        return C.c_void_p(a.ctypes.data)
    if hasattr(a, "data_ptr"):
        assert a.is_contiguous()
        return C.c_void_p(a.data_ptr())
    return C.c_void_p(int(a))


class Context:
    """thin RAII wrapper over dsss_ctx"""

    def __init__(self, max_frames, device=0):
        self.L = lib()
        h = C.c_void_p()
        rc = self.L.dsss_create(int(device), int(max_frames), C.byref(h))
        if rc != 0:
            raise DsssError("dsss_create: %s" % self.L.dsss_strerror(rc).decode())
        self.h = h
        self.max_frames = max_frames

    def close(self):
        if getattr(self, "h", None):
            self.L.dsss_destroy(self.h)
            self.h = None
        for k in ("_keep", "_pinned"):      # nothing of the caller's stays referenced by a closed context
            self.__dict__.pop(k, None)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc, what):
        if rc != 0:
            raise DsssError("%s: %s (%s)" % (what, self.L.dsss_strerror(rc).decode(),
                                             self.L.dsss_last_error(self.h).decode()))

    # ---- params
    def default_params(self):
        mp, op, mt, pg = MaskParams(), OrbParams(), MatchParams(), PGParams()
        self.L.dsss_mask_params_default(C.byref(mp)); self.L.dsss_orb_params_default(C.byref(op))
        self.L.dsss_match_params_default(C.byref(mt)); self.L.dsss_pg_params_default(C.byref(pg))
        return mp, op, mt, pg

    def set_params(self, mask=None, orb=None, match=None, pg=None):
        self._chk(self.L.dsss_set_params(self.h, C.byref(mask) if mask else None, C.byref(orb) if orb else None,
                                         C.byref(match) if match else None, C.byref(pg) if pg else None), "dsss_set_params")

    def sync(self):
        self._chk(self.L.dsss_sync(self.h), "dsss_sync")

    # ---- ranks (one process per GPU)
    def comm_unique_id(self):
        buf = np.zeros(128, np.uint8)
        rc = self.L.dsss_comm_unique_id(_ptr(buf))
        if rc != 0:
            raise DsssError("dsss_comm_unique_id: %s" % self.L.dsss_strerror(rc).decode())
        return buf

    def comm_init_rccl(self, uid128, rank, world):
        uid128 = np.ascontiguousarray(uid128, np.uint8); assert uid128.size == 128
        self._chk(self.L.dsss_comm_init(self.h, _ptr(uid128), int(rank), int(world)), "dsss_comm_init")

    def comm_init_callback(self, rank, world, fn):
        """fn(op, array): op 0 = sum the float64 array over the ranks in place, op 1 = all-gather: array is (world, n) uint8
        with this rank's row filled in"""
        def _cb(user, op, buf, n):
            try:
                if op == 0:
                    fn(0, np.ctypeslib.as_array(C.cast(buf, C.POINTER(C.c_double)), shape=(n,)))
                else:
                    fn(1, np.ctypeslib.as_array(C.cast(buf, C.POINTER(C.c_uint8)), shape=(world, n)))
                return 0
            except Exception as ex:                      # never let an exception cross the C boundary
                import traceback; traceback.print_exc()
                return 1
        self._cb_keep = COMM_FN(_cb)
        self._chk(self.L.dsss_comm_init_callback(self.h, int(rank), int(world), self._cb_keep, None), "dsss_comm_init_callback")

    def comm_init_device_callback(self, rank, world, fn):
        """fn(op, dev_ptr, n, stream): the library's collectives handed over as DEVICE buffers (dsss_comm_init_device_callback); op 0:
        n doubles to be summed in place, op 1: world x n bytes with this rank's slice in place; work must be ordered on `stream`"""
        def _cb(user, op, buf, n, stream):
            try:
                fn(int(op), int(buf or 0), int(n), int(stream or 0))
                return 0
            except Exception:
                import traceback; traceback.print_exc()
                return 1
        self._dcb_keep = COMM_DEV_FN(_cb)
        self._chk(self.L.dsss_comm_init_device_callback(self.h, int(rank), int(world), self._dcb_keep, None), "dsss_comm_init_device_callback")

    def comm_destroy(self):
        self._chk(self.L.dsss_comm_destroy(self.h), "dsss_comm_destroy")

    def comm_stats(self):
        r = C.c_int(); w = C.c_int(); b = C.c_double(); n = C.c_int64()
        self._chk(self.L.dsss_comm_stats(self.h, C.byref(r), C.byref(w), C.byref(b), C.byref(n)), "dsss_comm_stats")
        return r.value, w.value, b.value, n.value

    def frame_owner(self, nframes, frame):
        return self.L.dsss_comm_frame_owner(self.h, int(nframes), int(frame))

    def features_allgather(self, nframes):
        self._chk(self.L.dsss_features_allgather(self.h, int(nframes)), "dsss_features_allgather")

    def set_pg_partitions(self, nparts):
        self._chk(self.L.dsss_set_pg_partitions(self.h, int(nparts)), "dsss_set_pg_partitions")

    # ---- frames
    def frame_set(self, fid, raw, N, M, pose6, alt, gr):
        pose6 = np.ascontiguousarray(pose6, np.float64); alt = np.ascontiguousarray(alt, np.float64)
        gr = np.ascontiguousarray(gr, np.float64)
        assert pose6.shape == (N, 6) and alt.shape == (N,) and gr.shape == (M // 2,)
        if isinstance(raw, np.ndarray):
            raw = np.ascontiguousarray(raw, np.float64); assert raw.shape == (N, M)
        self._keep = getattr(self, "_keep", {}); self._keep[fid] = raw      # keep device tensors alive
        self._chk(self.L.dsss_frame_set(self.h, fid, _ptr(raw), N, M, _ptr(pose6), _ptr(alt), _ptr(gr)), "dsss_frame_set")

    def frames_args(self, ids, raws, Ns, Ms, poses, alts, grs):
        """The ARGUMENT ARRAYS of dsss_frames_set (ids, sizes, four arrays of per-frame pointers) built and validated once: what a C++
        caller of the C ABI simply holds.  Host arrays must be float64 and C-contiguous, raws may hold None, host arrays or CUDA tensors
        (float64, contiguous).  The object keeps every array alive; frames_set(args) is then the bare C call."""
        return FramesArgs(ids, raws, Ns, Ms, poses, alts, grs)

    def frames_set(self, ids, raws=None, Ns=None, Ms=None, poses=None, alts=None, grs=None):
        """dsss_frames_set: one call for many frames.  Either the seven per-frame lists, or one FramesArgs built by frames_args()."""
        a = ids if isinstance(ids, FramesArgs) else FramesArgs(ids, raws, Ns, Ms, poses, alts, grs)
        self._keep = getattr(self, "_keep", {})
        self._keep.update(a.keep)                              # the library borrows device images until the frame is set again
        self._chk(self.L.dsss_frames_set(self.h, a.n, _ptr(a.ids), _ptr(a.p_raw), _ptr(a.N), _ptr(a.M), _ptr(a.p_pose), _ptr(a.p_alt), _ptr(a.p_gr)), "dsss_frames_set")

    def extract(self, fid):
        n = C.c_int(0)
        self._chk(self.L.dsss_extract(self.h, fid, C.byref(n)), "dsss_extract")
        return n.value

    def extract_many(self, ids):
        ids = np.ascontiguousarray(ids, np.int32)
        self._chk(self.L.dsss_extract_many(self.h, _ptr(ids), len(ids)), "dsss_extract_many")

    def frame_norm(self, fid, N, M):
        norm = np.zeros((N, M), np.uint8); mask = np.zeros((N, M), np.uint8)
        self._chk(self.L.dsss_frame_get_norm(self.h, fid, _ptr(norm), _ptr(mask)), "dsss_frame_get_norm")
        return norm, mask

    def frame_level(self, fid, level, cap_rows, cap_cols):
        img = np.zeros(cap_rows * cap_cols, np.uint8); r = C.c_int(0); c = C.c_int(0)
        self._chk(self.L.dsss_frame_get_level(self.h, fid, level, _ptr(img), C.byref(r), C.byref(c)), "dsss_frame_get_level")
        return img[:r.value * c.value].reshape(r.value, c.value).copy()

    def frame_candidates(self, fid, level, cap=200000):
        x = np.zeros(cap, np.float32); y = np.zeros(cap, np.float32); r = np.zeros(cap, np.float32); n = C.c_int(0)
        self._chk(self.L.dsss_frame_get_candidates(self.h, fid, level, _ptr(x), _ptr(y), _ptr(r), cap, C.byref(n)), "dsss_frame_get_candidates")
        return x[:n.value].copy(), y[:n.value].copy(), r[:n.value].copy()

    def features_get(self, fid, cap=16384):
        kps = np.zeros(cap, KP_DTYPE); desc = np.zeros((cap, 32), np.uint8); geo = np.zeros((cap, 2), np.float64)
        n = C.c_int(0)
        self._chk(self.L.dsss_features_get(self.h, fid, _ptr(kps), _ptr(desc), _ptr(geo), cap, C.byref(n)), "dsss_features_get")
        return kps[:n.value].copy(), desc[:n.value].copy(), geo[:n.value].copy()

    def features_get_sift(self, fid, cap=16384):
        """Frame::dst of the SIFT call site: n x 128 float32 (integer-valued 0..255)"""
        d = np.zeros((cap, 128), np.float32); n = C.c_int(0)
        self._chk(self.L.dsss_features_get_sift(self.h, fid, _ptr(d), cap, C.byref(n)), "dsss_features_get_sift")
        return d[:n.value].copy()

    def features_set_sift(self, fid, d128):
        d = np.ascontiguousarray(d128, np.float32).reshape(-1, 128)
        self._chk(self.L.dsss_features_set_sift(self.h, fid, _ptr(d), len(d)), "dsss_features_set_sift")

    def features_set(self, fid, N, M, kps, desc, geo=None, bbox=None):
        kps = np.ascontiguousarray(kps, KP_DTYPE); desc = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32)
        if geo is not None: geo = np.ascontiguousarray(geo, np.float64).reshape(-1, 2)
        if bbox is not None: bbox = np.ascontiguousarray(bbox, np.float64)
        self._chk(self.L.dsss_features_set(self.h, fid, N, M, _ptr(kps), _ptr(desc), _ptr(geo), _ptr(bbox), len(kps)), "dsss_features_set")

    def frame_bbox(self, fid):
        bb = np.zeros(4, np.float64)
        self._chk(self.L.dsss_frame_bbox(self.h, fid, _ptr(bb)), "dsss_frame_bbox")
        return bb

    def frame_geo(self, fid, N, M):
        """Frame::geo_img in full (dsss_frame_get_geo): two N x M float64 arrays (x, y)"""
        gx = np.empty((N, M), np.float64); gy = np.empty((N, M), np.float64)
        self._chk(self.L.dsss_frame_get_geo(self.h, fid, _ptr(gx), _ptr(gy)), "dsss_frame_get_geo")
        return gx, gy

    def overlap(self, a, b):
        v = C.c_float(0)
        self._chk(self.L.dsss_overlap(self.h, a, b, C.byref(v)), "dsss_overlap")
        return v.value

    def pack_bytes(self):
        return int(self.L.dsss_features_pack_bytes(self.h))

    def features_pack(self, fid, buf):
        self._chk(self.L.dsss_features_pack(self.h, fid, _ptr(buf)), "dsss_features_pack")

    def features_unpack(self, fid, buf):
        self._chk(self.L.dsss_features_unpack(self.h, fid, _ptr(buf)), "dsss_features_unpack")

    # ---- matcher
    def match_pairs(self, src, tgt):
        src = np.ascontiguousarray(src, np.int32); tgt = np.ascontiguousarray(tgt, np.int32)
        self._chk(self.L.dsss_match_pairs(self.h, _ptr(src), _ptr(tgt), len(src)), "dsss_match_pairs")

    def match_dir(self, pair, d, cap=16384):
        nn = np.zeros(cap, np.int32); co = np.zeros(cap, np.int32)
        hist = C.c_int(0); cnt = C.c_int(0); model = C.c_double(0)
        self._chk(self.L.dsss_match_get_dir(self.h, pair, d, _ptr(nn), _ptr(co), cap, C.byref(hist), C.byref(cnt), C.byref(model)), "dsss_match_get_dir")
        return nn, co, hist.value, cnt.value, model.value

    def pair_is_active(self, pair):
        """False for a pair whose geo bounding boxes are disjoint (provably no match; skipped before any kernel runs)"""
        a = C.c_int(0)
        self._chk(self.L.dsss_match_pair_active(self.h, int(pair), C.byref(a)), "dsss_match_pair_active")
        return bool(a.value)

    def match_rows(self, pair):
        n = C.c_int(0)
        self._chk(self.L.dsss_match_get_rows(self.h, pair, None, 0, C.byref(n)), "dsss_match_get_rows")
        rows = np.zeros((max(n.value, 1), 6), np.float64)
        self._chk(self.L.dsss_match_get_rows(self.h, pair, _ptr(rows), len(rows), C.byref(n)), "dsss_match_get_rows")
        return rows[:n.value].copy()

    def match_kp7(self, pair):
        n = C.c_int(0)
        self._chk(self.L.dsss_match_get_kp7(self.h, pair, None, 0, C.byref(n)), "dsss_match_get_kp7")
        out = np.zeros((max(n.value, 1), 7), np.float64)
        self._chk(self.L.dsss_match_get_kp7(self.h, pair, _ptr(out), len(out), C.byref(n)), "dsss_match_get_kp7")
        return out[:n.value].copy()

    def match_total(self):
        a = C.c_int(0); b = C.c_int(0)
        self._chk(self.L.dsss_match_total(self.h, C.byref(a), C.byref(b)), "dsss_match_total")
        return a.value, b.value

    def descriptor_distance(self, fa, ia, fb, ib):
        d = C.c_int(0)
        self._chk(self.L.dsss_descriptor_distance(self.h, fa, ia, fb, ib, C.byref(d)), "dsss_descriptor_distance")
        return d.value

    # ---- optimizer
    def lc_solve_all(self):
        self._chk(self.L.dsss_lc_solve_all(self.h), "dsss_lc_solve_all")

    def lc_get(self, pair):
        n = C.c_int(0)
        self._chk(self.L.dsss_lc_get(self.h, pair, None, 0, C.byref(n)), "dsss_lc_get")
        out = np.zeros(max(n.value, 1), LC_DTYPE)
        self._chk(self.L.dsss_lc_get(self.h, pair, _ptr(out), len(out), C.byref(n)), "dsss_lc_get")
        return out[:n.value].copy()

    def lc_solve(self, id_s, id_t, kp7):
        kp7 = np.ascontiguousarray(kp7, np.float64).reshape(-1, 7)
        out = np.zeros(max(len(kp7), 1), LC_DTYPE)
        self._chk(self.L.dsss_lc_solve(self.h, id_s, id_t, _ptr(kp7), len(kp7), _ptr(out)), "dsss_lc_solve")
        return out[:len(kp7)].copy()

    def lc_solve_pairs(self, src, tgt, kp7_list):
        """LoopClosingTFs of many pairs in one launch from caller-built kp7 lists (dsss_lc_solve_pairs)"""
        src = np.ascontiguousarray(src, np.int32); tgt = np.ascontiguousarray(tgt, np.int32)
        off = np.zeros(len(src) + 1, np.int32)
        for p, k in enumerate(kp7_list):
            off[p + 1] = off[p] + len(k)
        kp7 = np.ascontiguousarray(np.concatenate([np.asarray(k, np.float64).reshape(-1, 7) for k in kp7_list]) if len(kp7_list) else np.zeros((0, 7)))
        self._chk(self.L.dsss_lc_solve_pairs(self.h, _ptr(src), _ptr(tgt), len(src), _ptr(kp7) if len(kp7) else None, _ptr(off)), "dsss_lc_solve_pairs")

    def triangulate(self, id_s, id_t, kp7):
        kp7 = np.ascontiguousarray(kp7, np.float64).reshape(-1, 7)
        out = np.zeros((max(len(kp7), 1), 7))
        self._chk(self.L.dsss_triangulate(self.h, id_s, id_t, _ptr(kp7), len(kp7), _ptr(out)), "dsss_triangulate")
        return out[:len(kp7)].copy()

    def triangulate_poses(self, kp7, in27):
        kp7 = np.ascontiguousarray(kp7, np.float64).reshape(-1, 7); in27 = np.ascontiguousarray(in27, np.float64).reshape(-1, 27)
        out = np.zeros((max(len(kp7), 1), 7))
        self._chk(self.L.dsss_triangulate_poses(self.h, _ptr(kp7), _ptr(in27), len(kp7), _ptr(out)), "dsss_triangulate_poses")
        return out[:len(kp7)].copy()

    def posegraph_select(self, nframes, cap=1 << 20):
        edges = np.zeros(cap, LCEDGE_DTYPE); n = C.c_int(0)
        self._chk(self.L.dsss_posegraph_select(self.h, nframes, _ptr(edges), cap, C.byref(n)), "dsss_posegraph_select")
        return edges[:n.value].copy()

    def _pinned_f64(self, key, shape):
        """page-locked output buffer (device-to-host copies into it run at PCIe speed), reused by the next call with
        the same key and shape: copy the result if it has to outlive the next solve"""
        cache = self.__dict__.setdefault("_pinned", {})
        buf = cache.get((key, shape))
        if buf is None:
            try:
                import torch
                buf = torch.empty(shape, dtype=torch.float64, pin_memory=True).numpy()
            except Exception:
                buf = np.empty(shape, np.float64)
            cache[(key, shape)] = buf
        return buf

    def posegraph_solve(self, nframes, total, want_rpy=True, pinned=False):
        """poses: total x 12 (R row-major, t); rpy: the reference's trajectory rows (roll pitch yaw x y z) or None.
        pinned=True returns views of page-locked buffers owned by this Context (overwritten by its next solve)."""
        poses = self._pinned_f64("poses", (total, 12)) if pinned else np.empty((total, 12), np.float64)
        stats = np.zeros(4, np.float64)
        rpy = np.empty((total, 6), np.float64) if want_rpy else None
        self._chk(self.L.dsss_posegraph_solve(self.h, nframes, _ptr(poses), _ptr(rpy) if want_rpy else None, _ptr(stats)), "dsss_posegraph_solve")
        return poses, rpy, stats

    def posegraph_update(self, nframes, total, want_rpy=False):
        """online use (dsss_posegraph_update): frames 0..nframes-1, started from the previous update's estimate, on the accumulated
        loop closures; consumes the LC result set the context holds.  Returns poses (total x 12), rpy or None, stats."""
        poses = np.empty((total, 12), np.float64); stats = np.zeros(4, np.float64)
        rpy = np.empty((total, 6), np.float64) if want_rpy else None
        self._chk(self.L.dsss_posegraph_update(self.h, nframes, _ptr(poses), _ptr(rpy) if want_rpy else None, _ptr(stats)), "dsss_posegraph_update")
        return poses, rpy, stats

    def posegraph_update_window(self, nframes, total, window, want_poses=True):
        """the incremental form (dsss_posegraph_update_window): only the last `window` frames are solved, conditioned on the frozen estimate
        of everything before them.  Returns (poses of ALL pings or None, stats of the window's LM)."""
        poses = np.empty((total, 12), np.float64) if want_poses else None
        stats = np.zeros(4, np.float64)
        self._chk(self.L.dsss_posegraph_update_window(self.h, nframes, int(window), _ptr(poses) if want_poses else None, None, _ptr(stats)), "dsss_posegraph_update_window")
        return poses, stats

    def posegraph_reset(self):
        self._chk(self.L.dsss_posegraph_reset(self.h), "dsss_posegraph_reset")

    def posegraph_online_edges(self):
        return int(self.L.dsss_posegraph_online_edges(self.h))

    def posegraph_solve_edges(self, dr6, edges):
        dr6 = np.ascontiguousarray(dr6, np.float64).reshape(-1, 6)
        edges = np.ascontiguousarray(edges, LCEDGE_DTYPE)
        poses = np.zeros((len(dr6), 12), np.float64); stats = np.zeros(4, np.float64)
        self._chk(self.L.dsss_posegraph_solve_edges(self.h, _ptr(dr6), len(dr6), _ptr(edges), len(edges), _ptr(poses), _ptr(stats)),
                  "dsss_posegraph_solve_edges")
        return poses, stats

    def posegraph_schedule(self):
        """panel levels of the last solve: (levels[n][4] = items, widest panel columns, tallest rows below, interface flag; trials)"""
        n = C.c_int(0); t = C.c_int(0)
        self._chk(self.L.dsss_posegraph_schedule_get(self.h, None, 0, C.byref(n), C.byref(t)), "dsss_posegraph_schedule_get")
        lv = np.zeros((max(n.value, 1), 4), np.int32)
        self._chk(self.L.dsss_posegraph_schedule_get(self.h, _ptr(lv), n.value, C.byref(n), C.byref(t)), "dsss_posegraph_schedule_get")
        return lv[:n.value], t.value

    # ---- instrumentation
    def profile(self, on=True):
        self._chk(self.L.dsss_profile_enable(self.h, 1 if on else 0), "dsss_profile_enable")

    def profile_reset(self):
        self._chk(self.L.dsss_profile_reset(self.h), "dsss_profile_reset")

    def profile_get(self):
        K = len(K_NAMES)                                   # == DSSS_K_COUNT (include/dsss.h)
        ms = np.zeros(K, np.float64); n = np.zeros(K, np.int64); wk = np.zeros(K, np.float64)
        self._chk(self.L.dsss_profile_get(self.h, _ptr(ms), _ptr(n)), "dsss_profile_get")
        self._chk(self.L.dsss_profile_get_work(self.h, _ptr(wk)), "dsss_profile_get_work")
        return {K_NAMES[i]: (float(ms[i]), int(n[i]), float(wk[i])) for i in range(K) if not K_NAMES[i].startswith("k2")}
