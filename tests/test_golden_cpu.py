"""Golden fixtures (tests/golden/*.npz, written by tools/make_golden.py with the oracle): the oracle must keep
reproducing them.  The reference has no vectors of its own, so these files are the pinned state of the
CPU restatement; tests/test_gpu_golden.py holds the HIP path to the same files."""
import os

import numpy as np

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _load(name):
    return np.load(os.path.join(GOLD, name))


def test_frame_fixture(orc):
    g = _load("frame_256x192.npz")
    mp = orc.mask_params()
    mp.factor, mp.width, mp.r, mp.side = float(g["mask_params"][0]), int(g["mask_params"][1]), int(g["mask_params"][2]), int(g["mask_params"][3])
    kps, desc, norm, msk = orc.detect_feature(g["raw"], mparams=mp)
    assert (norm == g["norm"]).all() and (msk == g["mask"]).all()
    assert len(kps) == len(g["kps"]) and kps.tobytes() == g["kps"].tobytes()          # keypoints bit for bit, same order
    assert (desc == g["desc"]).all()


def test_match_fixture(orc):
    g = _load("match_300x300.npz")
    N, M = int(g["N"]), int(g["M"]); i0, i1 = (int(v) for v in g["ids"])
    fr = []
    for f in range(2):
        pose, gr, kps = g["pose%d" % f], g["gr%d" % f], g["kps%d" % f]
        fr.append(dict(pose=pose, alt=g["alt%d" % f], gr=gr, kps=kps, desc=g["desc%d" % f], geo=orc.geo_at_kps(pose, gr, M, kps), bb=orc.geo_bbox(pose, gr, M)))
    a, b = fr
    d01 = orc.match_dir(i0, i1, N, a["kps"], a["desc"], a["geo"], b["kps"], b["desc"], b["geo"], b["bb"])
    d10 = orc.match_dir(i1, i0, N, b["kps"], b["desc"], b["geo"], a["kps"], a["desc"], a["geo"], a["bb"])
    assert (d01["nn"] == g["nn01"]).all() and (d01["corres"] == g["corres01"]).all()
    assert (d10["nn"] == g["nn10"]).all() and (d10["corres"] == g["corres10"]).all()
    rows = orc.robust_matching(i0, i1, N, N, a["kps"], a["desc"], a["geo"], a["bb"], b["kps"], b["desc"], b["geo"], b["bb"])
    assert rows.shape == g["rows"].shape and (rows == g["rows"]).all()
    kp7 = orc.get_kps_pairs(rows, i1, a["alt"], a["gr"], b["alt"], b["gr"])
    assert kp7.shape == g["kp7"].shape and (kp7 == g["kp7"]).all()


def test_lc_fixture(orc):
    g = _load("lc_32.npz")
    M = int(g["M"])
    lcs = orc.lc_solve(g["kp7"], g["pose0"], g["alt0"], g["gr0"], M, g["pose1"], g["alt1"], g["gr1"], M)
    ref = g["lcs"]
    assert (lcs["iters"] == ref["iters"]).all()
    assert np.allclose(lcs["rel"], ref["rel"], rtol=0, atol=1e-12)                    # same code, same libm: 1e-12
    assert np.allclose(lcs["var"], ref["var"], rtol=1e-10, atol=0)
    assert np.allclose(lcs["score"], ref["score"], rtol=0, atol=1e-10)


def test_posegraph_fixture(orc):
    g = _load("posegraph_3x64.npz")
    p = orc.pg_params(); p.add_noise = 0
    out, stats = orc.pg_solve(g["dr"], g["edges"], p)
    assert stats[0] == g["stats"][0] and np.isclose(stats[2], g["stats"][2], rtol=1e-10)
    assert np.abs(out - g["poses"]).max() < 1e-10
    out, stats = orc.pg_solve(g["dr"], g["edges"])                                    # default: initial-value noise on (optimizer.cpp:154-158)
    assert stats[0] == g["stats_default"][0]
    assert np.abs(out - g["poses_default"]).max() < 1e-10
