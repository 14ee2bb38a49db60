"""GPU parity against the committed golden fixtures (tests/golden/*.npz, see tools/make_golden.py): the HIP path
through the C ABI must reproduce the pinned oracle outputs without the oracle being present at run time."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _load(name):
    return np.load(os.path.join(GOLD, name))


@pytest.fixture()
def ctx():
    from diasss_amd import capi
    c = capi.Context(max_frames=4)
    yield c
    c.close()


def test_frame_fixture(ctx):
    """Frame ctor path: normalise, mask, pyramid, FAST, quadtree, orientation, rBRIEF, mask filter -- bit-exact"""
    g = _load("frame_256x192.npz")
    N, M = g["raw"].shape
    mp, _, _, _ = ctx.default_params()
    mp.factor, mp.width, mp.r, mp.side = float(g["mask_params"][0]), int(g["mask_params"][1]), int(g["mask_params"][2]), int(g["mask_params"][3])
    ctx.set_params(mask=mp)
    pose = np.zeros((N, 6)); pose[:, 3] = 0.05 * np.arange(N)
    ctx.frame_set(0, g["raw"], N, M, pose, np.full(N, 9.0), 0.05 * np.arange(M // 2, dtype=np.float64))
    n = ctx.extract(0)
    norm, msk = ctx.frame_norm(0, N, M)
    assert (norm == g["norm"]).all() and (msk == g["mask"]).all()
    kps, desc, _ = ctx.features_get(0)
    assert n == len(g["kps"]) and kps.tobytes() == g["kps"].tobytes()
    assert (desc == g["desc"]).all()


def test_match_fixture(ctx):
    """FEAmatcher::RobustMatching + GetKpsPairs on 300 x 300 descriptors: CorresID, rows and Vector7 -- bit-exact"""
    g = _load("match_300x300.npz")
    N, M = int(g["N"]), int(g["M"]); i0, i1 = (int(v) for v in g["ids"])
    for f, fid in enumerate((i0, i1)):
        ctx.frame_set(fid, None, N, M, g["pose%d" % f], g["alt%d" % f], g["gr%d" % f])
        ctx.features_set(fid, N, M, g["kps%d" % f], g["desc%d" % f])
    ctx.match_pairs([i0], [i1])
    n = len(g["kps0"])
    nn, co, _, _, _ = ctx.match_dir(0, 0)
    assert (nn[:n] == g["nn01"]).all() and (co[:n] == g["corres01"]).all()
    nn, co, _, _, _ = ctx.match_dir(0, 1)
    assert (nn[:n] == g["nn10"]).all() and (co[:n] == g["corres10"]).all()
    rows = ctx.match_rows(0)
    assert rows.shape == g["rows"].shape and (rows == g["rows"]).all()
    kp7 = ctx.match_kp7(0)
    assert kp7.shape == g["kp7"].shape and (kp7 == g["kp7"]).all()


def test_lc_fixture(ctx):
    """Optimizer::LoopClosingTFs on 32 Vector7: same LM path, relative pose within 1e-9"""
    g = _load("lc_32.npz")
    N, M = int(g["N"]), int(g["M"])
    for f, fid in enumerate((0, 2)):
        ctx.frame_set(fid, None, N, M, g["pose%d" % f], g["alt%d" % f], g["gr%d" % f])
    lcs = ctx.lc_solve(0, 2, g["kp7"])
    ref = g["lcs"]
    assert (lcs["iters"] == ref["iters"]).all()
    assert np.allclose(lcs["rel"], ref["rel"], rtol=0, atol=1e-9)
    assert np.allclose(lcs["var"], ref["var"], rtol=1e-6, atol=1e-15)
    assert np.allclose(lcs["score"], ref["score"], rtol=0, atol=1e-6)


def test_posegraph_fixture(ctx):
    """pose-graph LM on 3 x 64 pings with 10 loop closures: north_star tolerance 1e-6 on the optimised poses"""
    g = _load("posegraph_3x64.npz")
    _, _, _, pg = ctx.default_params()
    pg.add_noise = 0
    ctx.set_params(pg=pg)
    out, stats = ctx.posegraph_solve_edges(g["dr"], g["edges"])
    assert stats[0] == g["stats"][0] and np.isclose(stats[2], g["stats"][2], rtol=1e-6)
    assert np.abs(out - g["poses"]).max() < 1e-6
    pg.add_noise = 1
    ctx.set_params(pg=pg)
    out, stats = ctx.posegraph_solve_edges(g["dr"], g["edges"])
    assert stats[0] == g["stats_default"][0]
    assert np.abs(out - g["poses_default"]).max() < 1e-6
