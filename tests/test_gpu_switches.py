"""GPU: every DSSS_* environment switch that selects another code path of the product is held to the default path's result.

Round 5 pruned the experiment surface to ten switches (DESIGN.md names each; round 6 added DSSS_PG_LOCAL -- tests/test_gpu_multirank.py -- and DSSS_EX_SCRATCH_MB): two print diagnostics (DSSS_PG_VERBOSE, DSSS_EX_VERBOSE), the
others are read per call, so one process can flip them.  Paths that run the SAME arithmetic in another arrangement (ordering on the host,
other thread counts, other upload batch, the all-pairs matcher) must reproduce the default's bits; knobs that change the ELIMINATION ORDER
(bin size, dissection leaf and both-axes threshold) must reproduce it to rounding, with the same LM path."""
import hashlib
import os
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SAME_BITS_PG = [("DSSS_SYM_THREADS", "1"), ("DSSS_SYM_THREADS", "5")]
SAME_OPTIMUM_PG = [("DSSS_PG_BIN_COST", "300"), ("DSSS_PG_BIN_COST", "1200"), ("DSSS_PG_ND_BOTH", "1000000"), ("DSSS_PG_LEAF", "12"),
                   ("DSSS_PG_ND_INDEX", "0"),       # coordinate medians only: the ordering of rounds 2 - 4 (no chain-order cut candidate)
                   ("DSSS_PG_PARTS_ANALYSIS", "0"), ("DSSS_PG_PARTS_ANALYSIS", "3"), ("DSSS_PG_PARTS_ANALYSIS", "16")]      # the analysis as ONE graph (rounds 1 - 5) / by 3 / by 16 parts (round 6; default: by the size of the graph)
SAME_BITS_EX = [("DSSS_EX_UPLOAD_BATCH", "1"), ("DSSS_EX_UPLOAD_BATCH", "3"), ("DSSS_FS_THREADS", "1"),
                ("DSSS_EX_SCRATCH_MB", "1"), ("DSSS_EX_SCRATCH_MB", "40")]      # batches of one frame / of a few (the default bound, 24 GB, holds all five)


class _env:
    def __init__(self, k, v): self.k, self.v = k, v
    def __enter__(self): self.old = os.environ.get(self.k); os.environ[self.k] = self.v
    def __exit__(self, *a):
        if self.old is None: os.environ.pop(self.k, None)
        else: os.environ[self.k] = self.old


def test_pose_graph_switches_reproduce_the_default():
    from diasss_amd import capi
    from tests.test_gpu_configs import _lawnmower_graph
    dr, gt, edges = _lawnmower_graph(10, 12000, 4000, seed=11)           # 120 k poses, 4 k loop closures: bins, fronts of several panels, fused and unfused levels
    c = capi.Context(max_frames=2)
    ref, sref = c.posegraph_solve_edges(dr, edges)
    href = hashlib.sha1(ref.tobytes()).hexdigest()
    assert sref[0] >= 3 and sref[2] < 1e-6 * sref[1]
    for k, v in SAME_BITS_PG:
        with _env(k, v):
            p, s = c.posegraph_solve_edges(dr, edges)
        assert hashlib.sha1(p.tobytes()).hexdigest() == href and (s == sref).all(), "%s=%s changes the result" % (k, v)
    for k, v in SAME_OPTIMUM_PG:
        with _env(k, v):
            p, s = c.posegraph_solve_edges(dr, edges)
        assert s[0] == sref[0] and abs(s[2] - sref[2]) <= 1e-9 * sref[2] and np.abs(p - ref).max() < 1e-8, "%s=%s: %g" % (k, v, np.abs(p - ref).max())
    for nparts in (4, 8):                                                # the partitioned layout on one rank (the interface is dissected too)
        c.set_pg_partitions(nparts)
        p, s = c.posegraph_solve_edges(dr, edges)
        assert s[0] == sref[0] and np.abs(p - ref).max() < 1e-8
    c.close()


def test_extraction_switches_reproduce_the_default():
    import torch
    from diasss_amd import capi
    from diasss_amd.synth import Survey
    F, N, M = 5, 900, 512
    sv = Survey(F, N, M, seed=123, device="cuda:0")
    dev = [sv.frame(f) for f in range(F)]
    host = [d.cpu().pin_memory() for d in dev]
    ins = [sv.inputs(f) for f in range(F)]

    def run(raws):
        c = capi.Context(max_frames=F)
        c.frames_set(list(range(F)), raws, [N] * F, [M] * F, [i[0] for i in ins], [i[1] for i in ins], [i[2] for i in ins])
        c.extract_many(list(range(F)))
        h = hashlib.sha1()
        for f in range(F):
            k, d, g = c.features_get(f)
            h.update(k.tobytes()); h.update(d.tobytes()); h.update(g.tobytes())
        c.close()
        return h.hexdigest()
    href = run(dev)
    assert run(host) == href                                                 # page-locked host frames, uploaded under the kernels
    for k, v in SAME_BITS_EX:
        with _env(k, v):
            assert run(dev) == href, "%s=%s changes the features (device-resident frames)" % (k, v)
            assert run(host) == href, "%s=%s changes the features (host-resident frames)" % (k, v)
    torch.cuda.synchronize()


def test_matcher_switch_reproduces_the_default():
    """DSSS_MT_GRID=0 (the all-pairs gate + Hamming kernel) against the geo grid: first-stage and final CorresID, rows and kp7 of every
    pair of a small survey, bit for bit"""
    from diasss_amd import capi
    from diasss_amd.synth import Survey
    F, N, M = 6, 900, 512
    sv = Survey(F, N, M, seed=77, device="cuda:0")
    c = capi.Context(max_frames=F)
    ins = [sv.inputs(f) for f in range(F)]
    c.frames_set(list(range(F)), [sv.frame(f) for f in range(F)], [N] * F, [M] * F, [i[0] for i in ins], [i[1] for i in ins], [i[2] for i in ins])
    c.extract_many(list(range(F)))
    src = [i for i in range(F) for j in range(i + 1, F)]; tgt = [j for i in range(F) for j in range(i + 1, F)]

    def run():
        c.match_pairs(src, tgt)
        h = hashlib.sha1(); rows = 0
        for p in range(len(src)):
            for d in (0, 1):
                nn, co, hist, cnt, model = c.match_dir(p, d)
                h.update(nn.tobytes()); h.update(co.tobytes()); h.update(np.array([hist, cnt], np.int64).tobytes()); h.update(np.float64(model).tobytes())
            r = c.match_rows(p); h.update(r.tobytes()); h.update(c.match_kp7(p).tobytes()); rows += len(r)
        return h.hexdigest(), rows
    href, rows = run()
    assert rows > 50
    with _env("DSSS_MT_GRID", "0"):
        assert run() == (href, rows)
    c.close()


def test_ordering_on_degenerate_geometry():
    """the nested dissection on separator coordinates chosen against it: every y equal (one bucket, one extent zero), duplicated loop closures,
    a graph below the both-axes threshold -- with and without the chain-order cut candidate: another elimination order, the same optimum"""
    from diasss_amd import capi
    from tests.test_gpu_configs import _lawnmower_graph
    c = capi.Context(max_frames=2)
    for legs, per, nlc, flat in ((4, 3000, 900, True), (6, 2000, 1500, False), (2, 400, 12, True)):
        dr, gt, edges = _lawnmower_graph(legs, per, nlc, seed=29)
        if flat:
            dr = dr.copy(); dr[:, 4] = 0.0                                 # all poses on one line: y extent exactly zero
        edges = np.concatenate([edges, edges[: max(1, len(edges) // 7)]])   # duplicated loop closures
        with _env("DSSS_PG_ND_INDEX", "0"):
            pd, sd = c.posegraph_solve_edges(dr, edges)
            pd2, sd2 = c.posegraph_solve_edges(dr, edges)
        assert hashlib.sha1(pd.tobytes()).hexdigest() == hashlib.sha1(pd2.tobytes()).hexdigest() and (sd == sd2).all()
        pc, sc = c.posegraph_solve_edges(dr, edges)                         # the default (chain-order cuts)
        assert sc[0] == sd[0] and np.abs(pc - pd).max() < 1e-8
    c.close()
