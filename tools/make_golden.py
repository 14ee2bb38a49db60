#!/usr/bin/env python3
"""Generate the golden fixtures of SURVEY.md section 8(c) under tests/golden/ with the CPU oracle.

The reference itself cannot run in this image (OpenCV / GTSAM are absent), so these vectors pin the ORACLE: they
are inputs plus the outputs oracle/liboracle.so produced when they were generated, small enough to commit.  The CPU
suite checks that the oracle still reproduces them; the GPU suite checks the HIP path against the same files.

    python tools/make_golden.py          # rewrites tests/golden/*.npz (deterministic: seeds below)

Fixtures (all seeded, numpy only):
    frame_256x192.npz     synthetic 256 x 192 f64 waterfall -> normalised image, mask, keypoints, descriptors
    match_300x300.npz     two frames of 300 keypoints / descriptors -> CorresID both directions, rows, kp7
    lc_32.npz             32 reprojected matches (Vector7) of one frame pair -> loop-closure tuples
    posegraph_3x64.npz    3 frames x 64 pings of DR poses + loop-closure edges -> optimised poses
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import binding as O          # noqa: E402
from tests import helpers as H           # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def synth_frame(N, M, seed):
    """seafloor of Gaussian blobs under Rayleigh speckle, like diasss_amd/synth.py but numpy only and tiny"""
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:N, 0:M].astype(np.float64)
    img = np.full((N, M), 0.35)
    for _ in range(60):
        cy, cx, s, a = rng.uniform(0, N), rng.uniform(0, M), rng.uniform(2.0, 7.0), rng.uniform(0.3, 1.2)
        img += a * np.exp(-((y - cy) ** 2 + (x - cx) ** 2) / (2 * s * s))
    img *= rng.rayleigh(0.8, (N, M)) + 0.2
    return np.ascontiguousarray(img)


def frame_fixture():
    N, M = 256, 192
    raw = synth_frame(N, M, 20240601)
    mp = O.mask_params(); mp.side = 24          # the default side margin (150 pings) would blank a 256-ping frame
    kps, desc, norm, msk = O.detect_feature(raw, mparams=mp)
    assert len(kps) > 50, "fixture frame yields too few keypoints (%d)" % len(kps)
    np.savez_compressed(os.path.join(OUT, "frame_256x192.npz"), raw=raw, norm=norm, mask=msk, kps=kps, desc=desc,
                        mask_params=np.array([mp.factor, mp.width, mp.r, mp.side], np.float64))
    return len(kps)


def match_fixture():
    """two passes over the same swath (frame ids 0 and 2: same heading), the second sees the first one's landmarks with a
    few descriptor bits flipped and a few pings of along-track jitter"""
    N, M, n = 700, 480, 300
    fr = []
    for f in range(2):
        pose, alt, gr = H.track(N, M, 0, seed=3)
        pose = pose.copy(); pose[:, 4] += 0.3 * f
        if f == 0:
            kps, desc = H.random_features(N, M, n, 50 + 10 * n)
        else:
            kps, desc = fr[0]["kps"].copy(), fr[0]["desc"].copy()
            rng = np.random.default_rng(n)
            for i in range(n):
                for bit in rng.choice(256, rng.integers(0, 20), replace=False):
                    desc[i, bit // 8] ^= np.uint8(1 << (bit % 8))
            kps["y"] += rng.integers(-3, 4, n).astype(np.float32)
        fr.append(dict(pose=pose, alt=alt, gr=gr, kps=kps, desc=desc, geo=O.geo_at_kps(pose, gr, M, kps), bb=O.geo_bbox(pose, gr, M)))
    a, b = fr
    d01 = O.match_dir(0, 2, N, a["kps"], a["desc"], a["geo"], b["kps"], b["desc"], b["geo"], b["bb"])
    d10 = O.match_dir(2, 0, N, b["kps"], b["desc"], b["geo"], a["kps"], a["desc"], a["geo"], a["bb"])
    rows = O.robust_matching(0, 2, N, N, a["kps"], a["desc"], a["geo"], a["bb"], b["kps"], b["desc"], b["geo"], b["bb"])
    kp7 = O.get_kps_pairs(rows, 2, a["alt"], a["gr"], b["alt"], b["gr"])
    assert len(kp7) >= 32, "fixture pair yields too few matches (%d rows, %d kp7)" % (len(rows), len(kp7))
    np.savez_compressed(os.path.join(OUT, "match_300x300.npz"), N=N, M=M, ids=np.array([0, 2]),
                        pose0=a["pose"], alt0=a["alt"], gr0=a["gr"], kps0=a["kps"], desc0=a["desc"],
                        pose1=b["pose"], alt1=b["alt"], gr1=b["gr"], kps1=b["kps"], desc1=b["desc"],
                        nn01=d01["nn"], corres01=d01["corres"], nn10=d10["nn"], corres10=d10["corres"], rows=rows, kp7=kp7)
    return fr, kp7, N, M


def lc_fixture(fr, kp7, N, M):
    a, b = fr
    kp7 = kp7[:32].copy()
    assert len(kp7) == 32
    lcs = O.lc_solve(kp7, a["pose"], a["alt"], a["gr"], M, b["pose"], b["alt"], b["gr"], M)
    np.savez_compressed(os.path.join(OUT, "lc_32.npz"), N=N, M=M, kp7=kp7,
                        pose0=a["pose"], alt0=a["alt"], gr0=a["gr"], pose1=b["pose"], alt1=b["alt"], gr1=b["gr"], lcs=lcs)
    return lcs


def posegraph_fixture():
    F, N = 3, 64
    rng = np.random.default_rng(11)
    dr = np.concatenate([H.track(N, 480, f, seed=9)[0] for f in range(F)])
    dr[:, 5] = 20.0 + 0.01 * rng.standard_normal(F * N).cumsum()
    edges = np.zeros(10, O.LCEDGE_DTYPE)
    for e in range(len(edges)):
        fa, fb = (0, 1) if e % 2 == 0 else (1, 2)
        ia, ib = fa * N + int(rng.integers(4, N - 4)), fb * N + int(rng.integers(4, N - 4))
        # measured relative pose = DR relative pose perturbed by a few centimetres / milliradians
        w = 0.002 * rng.standard_normal(3)
        th = np.linalg.norm(w); K = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
        R = np.eye(3) + np.sin(th) / th * K + (1 - np.cos(th)) / th ** 2 * K @ K

        def pose_R(p):
            v = p[:3]; t = np.linalg.norm(v)
            if t < 1e-12:
                return np.eye(3)
            Kx = np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]])
            return np.eye(3) + np.sin(t) / t * Kx + (1 - np.cos(t)) / t ** 2 * Kx @ Kx
        Ra, Rb = pose_R(dr[ia]), pose_R(dr[ib])
        Rrel = Ra.T @ Rb @ R
        trel = Ra.T @ (dr[ib, 3:] - dr[ia, 3:]) + 0.05 * rng.standard_normal(3)
        edges[e]["a"], edges[e]["b"] = ia, ib
        edges[e]["rel"][:9] = Rrel.reshape(-1); edges[e]["rel"][9:] = trel
        edges[e]["var"] = [1e-4, 1e-4, 1e-4, 1e-2, 1e-2, 1e-2]
    edges = edges[np.argsort(edges["b"], kind="stable")]
    p = O.pg_params(); p.add_noise = 0
    out, stats = O.pg_solve(dr, edges, p)
    p1 = O.pg_params()
    out_noise, stats_noise = O.pg_solve(dr, edges, p1)
    np.savez_compressed(os.path.join(OUT, "posegraph_3x64.npz"), dr=dr, edges=edges, poses=out, stats=stats,
                        poses_default=out_noise, stats_default=stats_noise)
    return stats


def main():
    os.makedirs(OUT, exist_ok=True)
    nk = frame_fixture()
    fr, kp7, N, M = match_fixture()
    lcs = lc_fixture(fr, kp7, N, M)
    st = posegraph_fixture()
    print("frame: %d keypoints; match: %d kp7 rows; lc: iters %s...; posegraph: %d LM iterations, error %.6g -> %.6g"
          % (nk, len(kp7), lcs["iters"][:6].tolist(), int(st[0]), st[1], st[2]))
    for f in sorted(os.listdir(OUT)):
        print("  %-24s %7.1f KB" % (f, os.path.getsize(os.path.join(OUT, f)) / 1024))


if __name__ == "__main__":
    main()
