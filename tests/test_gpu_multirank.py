"""GPU: the sharded pose-graph solve (contiguous frame blocks per rank, ONE all-reduce of the reduced Hessian on the interface
per LM trial, replicated interface solve) must reproduce the single-rank trajectory.
  - partitions on ONE rank: the interface machinery (rank-partitioned ordering, packed update matrices, interface fronts) with
    the all-reduce as a no-op;
  - two ranks sharing the one GPU of the test box over gloo (RCCL refuses two ranks on one device): the library's host-callback
    transport carries exactly the buffers the RCCL transport sums;
  - a one-rank RCCL communicator: ncclCommInitRank / ncclAllReduce / ncclAllGather really are called from libdsss.so."""
import os
import sys
import numpy as np
import pytest
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _survey(F=6):
    from diasss_amd.synth import Survey
    N, M = 700, 480
    sv = Survey(F, N, M, seed=91)
    raws = [sv.frame(f).numpy() for f in range(F)]
    poses = [sv.inputs(f)[0] for f in range(F)]; alts = [sv.inputs(f)[1] for f in range(F)]; grs = [sv.inputs(f)[2] for f in range(F)]
    return F, raws, poses, alts, grs


def _sift_mode(pipe):
    """the 128-float rows + the L2 matcher on them (DSSS_DESC_SIFT128, use_l2 = 2)"""
    from diasss_amd import capi
    mp_, op_, mt_, pg_ = pipe.ctx.default_params()
    op_.descriptor = capi.DESC_SIFT128; mt_.use_l2 = 2
    pipe.ctx.set_params(orb=op_, match=mt_)


def _reference(F, raws, poses, alts, grs, sift=False):
    from diasss_amd.pipeline import Pipeline
    pipe = Pipeline(F, device=0)
    if sift:
        _sift_mode(pipe)
    ref, ref_stats = pipe.run(raws, poses, alts, grs)
    ref = ref.copy()
    n_edges = len(pipe.ctx.posegraph_select(F))
    pipe.close()
    return ref, ref_stats, n_edges


@pytest.mark.parametrize("nparts", [2, 3, 6])
def test_partitions_on_one_rank_equal_unpartitioned(nparts):
    from diasss_amd.pipeline import Pipeline
    F, raws, poses, alts, grs = _survey()
    ref, ref_stats, n_edges = _reference(F, raws, poses, alts, grs)
    assert n_edges > 20
    pipe = Pipeline(F, device=0)
    pipe.ctx.set_pg_partitions(nparts)
    out, stats = pipe.run(raws, poses, alts, grs)
    assert stats[0] == ref_stats[0]
    assert np.abs(out - ref).max() < 1e-9
    pipe.close()


def test_partitioned_solve_edges_large_chain(orc):
    """30 000 poses, 400 loop closures, 1 / 4 / 8 partitions: same optimum; the stand-alone entry point partitions by pose blocks"""
    from diasss_amd import capi
    n = 30000
    rng = np.random.default_rng(3)
    dr = np.zeros((n, 6)); dr[:, 3] = 0.05 * np.arange(n); dr[:, 4] = 2.0 * np.sin(np.arange(n) / 500.0); dr[:, 2] = 0.05 * np.cos(np.arange(n) / 700.0)
    edges = np.zeros(400, orc.LCEDGE_DTYPE)
    b = np.sort(rng.choice(np.arange(2000, n), 400, replace=False)); a = (b - rng.integers(500, 1900, 400)).astype(np.int32)
    import ctypes as C
    for e in range(400):
        Ta = orc.Pose(); Tb = orc.Pose(); Tr = orc.Pose()
        orc.lib().orc_pose_from_rodrigues(orc.dp(np.ascontiguousarray(dr[a[e]])), C.byref(Ta))
        orc.lib().orc_pose_from_rodrigues(orc.dp(np.ascontiguousarray(dr[b[e]])), C.byref(Tb))
        orc.lib().orc_pose_between(C.byref(Ta), C.byref(Tb), C.byref(Tr))
        rel = np.concatenate([np.array(Tr.R), np.array(Tr.t)]); rel[9:] += rng.normal(0, 0.05, 3)
        edges["a"][e] = a[e]; edges["b"][e] = b[e]; edges["rel"][e] = rel; edges["var"][e] = [1e-5, 1e-5, 1e-4, 1e-2, 1e-2, 1e-2]
    c = capi.Context(max_frames=2)
    p1, s1 = c.posegraph_solve_edges(dr, edges)
    for nparts in (4, 8):
        c.set_pg_partitions(nparts)
        pk, sk = c.posegraph_solve_edges(dr, edges)
        assert sk[0] == s1[0] and np.abs(pk - p1).max() < 1e-9
        pk2, _ = c.posegraph_solve_edges(dr, edges)
        assert (pk2 == pk).all()                                   # bit-reproducible
    c.close()


def _worker(rank, world, port, q, backend):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    if backend == "nccl":
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    from diasss_amd.pipeline import Pipeline, shard_frames
    F, raws, poses, alts, grs = _survey()
    mine = set(shard_frames(F, rank, world))
    raws = [r if f in mine else None for f, r in enumerate(raws)]
    pipe = Pipeline(F, device=0, rank=rank, world=world, dist=dist, force_collectives=True)
    if os.environ.get("DSSS_TEST_SIFT") == "1":
        _sift_mode(pipe)
    if backend == "nccl":
        pipe.ctx.set_pg_partitions(3)                               # one RCCL rank, three partitions: the all-reduce runs on real buffers
    out, stats = pipe.run(raws, poses, alts, grs)
    out2, _ = pipe.run(raws, poses, alts, grs)
    q.put((rank, out.copy(), np.array(stats), bool((out2 == out).all()), pipe.ctx.comm_stats()))
    pipe.close()
    dist.destroy_process_group()


def _run(world, backend, port):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, backend)) for r in range(world)]
    for p in procs: p.start()
    res = sorted([q.get(timeout=600) for _ in range(world)], key=lambda t: t[0])
    for p in procs: p.join(timeout=60)
    return res


@pytest.mark.parametrize("world", [2, 3])
def test_ranks_over_gloo_equal_single_rank(world):
    F, raws, poses, alts, grs = _survey()
    ref, ref_stats, n_edges = _reference(F, raws, poses, alts, grs)
    res = _run(world, "gloo", 29600 + os.getpid() % 1000 + world)
    for rank, out, stats, repro, cs in res:
        assert stats[0] == ref_stats[0]
        assert np.abs(out - ref).max() < 1e-9
        assert repro                                                # run to run identical on every rank
        assert cs[1] == world and cs[3] > 0 and cs[2] > 0           # all-reduces happened, bytes counted
    for r in range(1, world):
        assert (res[0][1] == res[r][1]).all()                       # every rank ends with the same bits


def test_ranks_with_the_replicated_analysis(monkeypatch):
    """Round 6 made the RANK-LOCAL analysis the default on several ranks (a rank orders and analyses its own separators + the interface,
    which is one dense front summed in place; dsss_pg.hip).  The replicated analysis of rounds 2 - 5 -- every rank analyses the whole
    graph, the interface is a tree, update matrices are packed for the all-reduce -- stays as the fall-back for interfaces of more than
    PG_LOCAL_IFACE_MAX separators and is selected here by DSSS_PG_LOCAL=0 (the spawned ranks inherit it): same trajectory, fewer or more
    bytes summed."""
    F, raws, poses, alts, grs = _survey()
    ref, ref_stats, n_edges = _reference(F, raws, poses, alts, grs)
    loc = _run(2, "gloo", 29100 + os.getpid() % 400)
    monkeypatch.setenv("DSSS_PG_LOCAL", "0")
    rep = _run(2, "gloo", 29500 + os.getpid() % 400)
    for res in (loc, rep):
        for rank, out, stats, repro, cs in res:
            assert stats[0] == ref_stats[0] and np.abs(out - ref).max() < 1e-9 and repro
        assert (res[0][1] == res[1][1]).all()
    assert np.abs(loc[0][1] - rep[0][1]).max() < 1e-9
    assert loc[0][4][2] != rep[0][4][2]                            # (bytes through the collectives: the two modes sum different buffers)


def test_ranks_over_gloo_in_sift_mode(monkeypatch):
    """the sharded pipeline with DSSS_DESC_SIFT128: the feature all-gather carries the 128-byte rows, every rank matches its pairs by L2 on them"""
    F, raws, poses, alts, grs = _survey()
    ref, ref_stats, n_edges = _reference(F, raws, poses, alts, grs, sift=True)
    ref0, _, n0 = _reference(F, raws, poses, alts, grs)
    assert n_edges > 10 and (n_edges != n0 or np.abs(ref - ref0).max() > 1e-9)      # (another matcher: another set of loop closures)
    monkeypatch.setenv("DSSS_TEST_SIFT", "1")                      # (the spawned ranks inherit it)
    res = _run(2, "gloo", 29300 + os.getpid() % 500)
    for rank, out, stats, repro, cs in res:
        assert stats[0] == ref_stats[0] and np.abs(out - ref).max() < 1e-9 and repro
    assert (res[0][1] == res[1][1]).all()


def test_rccl_one_rank_communicator():
    F, raws, poses, alts, grs = _survey()
    ref, ref_stats, n_edges = _reference(F, raws, poses, alts, grs)
    (rank, out, stats, repro, cs), = _run(1, "nccl", 29700 + os.getpid() % 1000)
    assert stats[0] == ref_stats[0] and np.abs(out - ref).max() < 1e-9 and repro
    assert cs[1] == 1 and cs[3] > 0


# ---------------------------------------------------------------------------------------------------------------------
# BASELINE config 4 ("200 frames sharded 8 ways") as far as ONE GPU can take it: the full C3 survey (200 x 2000 x 1024, dense
# all-pairs) with the pose graph cut into the 8 partitions of the 8-GPU layout on one rank, and with 2 ranks (gloo: RCCL
# refuses two ranks on one device) that share the GPU -- feature all-gather, pairs to the owner of the target frame, edge
# exchange, partitioned solve with ONE interface all-reduce per LM trial -- against the single-rank trajectory.
C3 = dict(F=200, N=2000, M=1024, seed=20240601 + 1)


def _c3_inputs(mine):
    from diasss_amd.synth import Survey
    sv = Survey(C3["F"], C3["N"], C3["M"], seed=C3["seed"], device="cuda:0")
    raws = [sv.frame(f) if f in mine else None for f in range(C3["F"])]
    ins = [sv.inputs(f) for f in range(C3["F"])]
    return raws, [i[0] for i in ins], [i[1] for i in ins], [i[2] for i in ins]


def _c3_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from diasss_amd.pipeline import Pipeline, shard_frames
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    F = C3["F"]
    raws, poses, alts, grs = _c3_inputs(set(shard_frames(F, rank, world)))
    pipe = Pipeline(F, device=0, rank=rank, world=world, dist=dist)
    out, stats = pipe.run(raws, poses, alts, grs)
    torch.cuda.synchronize()
    q.put((rank, out.copy(), np.array(stats), pipe.ctx.comm_stats()))
    pipe.close()
    dist.destroy_process_group()


def test_config_C4_full_size_8_partitions_and_2_ranks(orc):
    from diasss_amd.pipeline import Pipeline
    F = C3["F"]
    raws, poses, alts, grs = _c3_inputs(set(range(F)))
    pipe = Pipeline(F, device=0)
    ref, ref_stats = pipe.run(raws, poses, alts, grs)
    ref = ref.copy(); ref_stats = np.array(ref_stats)
    g_edges = pipe.ctx.posegraph_select(F)
    n_edges = len(g_edges)
    # the ORACLE's LM on the same graph (its reduced system through a sparse LU, oracle/binding.py): every layout is held to IT (round 6),
    # at the default stopping rule and converged.  The selected edges are the oracle's own to 1e-9 (test_config_C3_full_size_every_stage_vs_oracle).
    o_edges = np.zeros(n_edges, orc.LCEDGE_DTYPE)
    for k in ("a", "b", "rel", "var"):
        o_edges[k] = g_edges[k]
    dr = np.concatenate(poses)
    o_def, o_def_stats = orc.pg_solve(dr, o_edges, solver="sparse")
    po_ = orc.pg_params(); po_.rel_tol = 1e-13; po_.abs_tol = 1e-13
    o_conv, o_conv_stats = orc.pg_solve(dr, o_edges, po_, solver="sparse")
    assert n_edges > 5000 and ref_stats[0] >= 3
    span = np.abs(ref[:, 9:]).max()                                # the track spans hundreds of metres
    pipe.ctx.set_pg_partitions(8)                                  # the 8-GPU layout of the solve, every partition on this rank
    p8, s8 = pipe.run(raws, poses, alts, grs)
    p8 = p8.copy()
    d8 = np.abs(p8 - ref).max()
    p8b, _ = pipe.run(raws, poses, alts, grs)
    assert (p8b == p8).all()                                       # bit-reproducible
    # The same two layouts run TO CONVERGENCE (stopping tolerances tightened: GTSAM's defaults stop an LM iterate about a micrometre short
    # of the optimum of a 4 km track, and the reduced system's conditioning -- prior sigma 1e-6 against loop-closure sigmas of decimetres --
    # lets two elimination orders stop that far apart): the optimum itself does not depend on the layout.
    mp_, op_, mt_, pg = pipe.ctx.default_params()
    pg.rel_tol = 1e-13; pg.abs_tol = 1e-13
    pipe.ctx.set_params(pg=pg)
    c8, _, cs8 = pipe.ctx.posegraph_solve(F, len(ref), want_rpy=False); c8 = c8.copy(); cs8 = np.array(cs8)
    pipe.ctx.set_pg_partitions(1)
    c1, _, cs1 = pipe.ctx.posegraph_solve(F, len(ref), want_rpy=False); c1 = c1.copy(); cs1 = np.array(cs1)
    dconv = np.abs(c8 - c1).max()
    pipe.close()
    del raws
    import torch; torch.cuda.empty_cache()
    res = _run_fn(_c3_worker, 2, 29900 + os.getpid() % 1000)
    d2 = max(np.abs(out - ref).max() for _, out, _, _ in res)
    print("C4 on one GPU: %d LC edges, %d LM iterations, track extent %.0f m; max |pose - single rank|: 8 partitions %.3g, 2 gloo ranks %.3g; "
          "all-reduce bytes per rank %.1f MB in %d calls" % (n_edges, ref_stats[0], span, d8, d2, res[0][3][2] / 1e6, res[0][3][3]))
    print("converged runs: %d / %d iterations, error %.9g / %.9g, max |8 partitions - 1| %.3g; default stopping rule: single rank to its converged run %.3g"
          % (cs8[0], cs1[0], cs8[2], cs1[2], dconv, np.abs(ref - c1).max()))
    to_oracle = dict(single=np.abs(ref - o_def).max(), parts8=np.abs(p8 - o_def).max(), ranks2=max(np.abs(out - o_def).max() for _, out, _, _ in res))
    conv_to_oracle = dict(single=np.abs(c1 - o_conv).max(), parts8=np.abs(c8 - o_conv).max())
    print("against the ORACLE's LM (%d iterations, objective %.12e): default stopping rule %s; converged (%d iterations) %s"
          % (o_def_stats[0], o_def_stats[2], {k: "%.3g" % v for k, v in to_oracle.items()}, o_conv_stats[0], {k: "%.3g" % v for k, v in conv_to_oracle.items()}))
    assert s8[0] == ref_stats[0] == o_def_stats[0] and abs(s8[2] - ref_stats[2]) <= 1e-6 * ref_stats[2]
    assert abs(ref_stats[2] - o_def_stats[2]) <= 1e-9 * o_def_stats[2]
    # (a) the OPTIMUM does not depend on the layout and is the oracle's: north_star's 1e-6, absolute, on a 4 km track (measured 4e-8 between
    #     layouts, 1.3e-7 .. 1.9e-7 to the oracle)
    assert dconv < 2e-7
    assert max(conv_to_oracle.values()) < 1e-6, conv_to_oracle
    # (b) GTSAM's default stopping rule returns an iterate ~0.1 m short of that optimum, where the rounding of any f64 elimination moves it
    #     by about a micrometre (the oracle against itself: 4e-7 .. 7e-7 between LU orderings / with iterative refinement, tools/pg_parity.py):
    #     layouts 1e-6 apart, 2e-6 .. 4e-6 from the oracle.  Not north_star's 1e-6 -- stated in DESIGN.md section 0, not hidden in a tolerance.
    assert d8 < 3e-6
    assert max(to_oracle.values()) < 1e-5, to_oracle
    for rank, out, stats, cs in res:
        assert stats[0] == ref_stats[0] and abs(stats[2] - ref_stats[2]) <= 1e-6 * ref_stats[2]
        assert np.abs(out - ref).max() < 3e-6
        assert cs[1] == 2 and cs[3] > 0 and cs[2] > 0
    assert (res[0][1] == res[1][1]).all()                          # identical bits on both ranks


def _c4_rccl_worker(rank, world, port, q):
    """one rank per PHYSICAL device over RCCL: the launch BASELINE config 4 names"""
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from diasss_amd.pipeline import Pipeline, shard_frames
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    F = C3["F"]
    from diasss_amd.synth import Survey
    sv = Survey(F, C3["N"], C3["M"], seed=C3["seed"], device="cuda:%d" % rank)
    mine = set(shard_frames(F, rank, world))
    raws = [sv.frame(f) if f in mine else None for f in range(F)]
    ins = [sv.inputs(f) for f in range(F)]
    pipe = Pipeline(F, device=rank, rank=rank, world=world, dist=dist)
    out, stats = pipe.run(raws, [i[0] for i in ins], [i[1] for i in ins], [i[2] for i in ins])
    out = out.copy()
    out2, _ = pipe.run(raws, [i[0] for i in ins], [i[1] for i in ins], [i[2] for i in ins])
    torch.cuda.synchronize()
    q.put((rank, out, np.array(stats), pipe.ctx.comm_stats(), bool((out2 == out).all())))
    pipe.close()
    dist.destroy_process_group()


def test_config_C4_rccl_one_rank_per_device():
    """BASELINE config 4 as it is written -- 200 frames sharded over min(device count, 8) ranks, ONE RANK PER DEVICE, the library's own
    RCCL communicator (all-gather of the features, edge exchange, one all-reduce of the reduced Hessian per LM trial over xGMI) --
    against the single-rank trajectory, with the assertions of the 2-rank gloo test above.  Skips on a box with fewer than two
    devices (the builder's and the round-end test box have one): it runs the day the suite meets a multi-GPU node.  The device
    count is read without initialising the GPU in this process; the ranks are spawned children."""
    import torch
    world = min(torch.cuda.device_count(), 8)
    if world < 2:
        pytest.skip("needs at least two devices (RCCL refuses two ranks on one device)")
    from diasss_amd.pipeline import Pipeline
    F = C3["F"]
    raws, poses, alts, grs = _c3_inputs(set(range(F)))
    pipe = Pipeline(F, device=0)
    ref, ref_stats = pipe.run(raws, poses, alts, grs)
    ref = ref.copy(); ref_stats = np.array(ref_stats)
    pipe.close()
    del raws
    torch.cuda.empty_cache()
    res = _run_fn(_c4_rccl_worker, world, 29100 + os.getpid() % 500)
    for rank, out, stats, cs, repro in res:
        assert stats[0] == ref_stats[0] and abs(stats[2] - ref_stats[2]) <= 1e-6 * ref_stats[2]
        assert np.abs(out - ref).max() < 1e-6
        assert cs[1] == world and cs[3] > 0 and cs[2] > 0 and repro
    for r in range(1, world):
        assert (res[0][1] == res[r][1]).all()                      # identical bits on every rank


def _run_fn(fn, world, port, timeout=900):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=fn, args=(r, world, port, q)) for r in range(world)]
    for p in procs: p.start()
    res = sorted([q.get(timeout=timeout) for _ in range(world)], key=lambda t: t[0])
    for p in procs: p.join(timeout=120)
    return res


# ---------------------------------------------------------------------------------------------------------------------
# dsss_posegraph_solve_edges takes loop closures in either direction (a > b as well as a < b).  A factor belongs to the rank of
# its HIGHER pose (the rule the analysis builds the interface on); an edge that crosses ranks with a > b used to be evaluated by
# the owner of b, against a pose it never updates.
def _reversed_edge_graph():
    from oracle import binding as orc
    import ctypes as C
    n = 24000
    rng = np.random.default_rng(11)
    dr = np.zeros((n, 6)); dr[:, 3] = 0.05 * np.arange(n); dr[:, 4] = 2.0 * np.sin(np.arange(n) / 500.0); dr[:, 2] = 0.05 * np.cos(np.arange(n) / 700.0)
    ne = 300
    hi = np.sort(rng.choice(np.arange(3000, n), ne, replace=False)); lo = (hi - rng.integers(500, 2900, ne)).astype(np.int64)
    flip = rng.random(ne) < 0.5                                   # half of the edges are given as (higher, lower)
    a = np.where(flip, hi, lo).astype(np.int32); b = np.where(flip, lo, hi).astype(np.int32)
    edges = np.zeros(ne, orc.LCEDGE_DTYPE)
    for e in range(ne):
        Ta = orc.Pose(); Tb = orc.Pose(); Tr = orc.Pose()
        orc.lib().orc_pose_from_rodrigues(orc.dp(np.ascontiguousarray(dr[a[e]])), C.byref(Ta))
        orc.lib().orc_pose_from_rodrigues(orc.dp(np.ascontiguousarray(dr[b[e]])), C.byref(Tb))
        orc.lib().orc_pose_between(C.byref(Ta), C.byref(Tb), C.byref(Tr))
        rel = np.concatenate([np.array(Tr.R), np.array(Tr.t)]); rel[9:] += rng.normal(0, 0.05, 3)
        edges["a"][e] = a[e]; edges["b"][e] = b[e]; edges["rel"][e] = rel; edges["var"][e] = [1e-5, 1e-5, 1e-4, 1e-2, 1e-2, 1e-2]
    assert ((a > b) & ((a >= n // 2) != (b >= n // 2))).sum() > 5   # reversed edges that cross the 2-rank cut exist
    return dr, edges


def _rev_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from diasss_amd import capi
    from diasss_amd.pipeline import make_comm
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dr, edges = _reversed_edge_graph()
    c = capi.Context(max_frames=2)
    make_comm(c, dist, rank, world)
    p, s = c.posegraph_solve_edges(dr, edges)
    q.put((rank, p.copy(), np.array(s)))
    c.close()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_reversed_edges_across_ranks(orc, world):
    from diasss_amd import capi
    dr, edges = _reversed_edge_graph()
    c = capi.Context(max_frames=2)
    ref, sref = c.posegraph_solve_edges(dr, edges)
    c.close()
    o_poses, o_stats = orc.pg_solve(dr, edges)                     # the oracle takes either direction as well
    assert sref[0] == o_stats[0] and np.abs(ref - o_poses).max() < 1e-6
    res = _run_fn(_rev_worker, world, 29300 + os.getpid() % 500 + world)
    for rank, p, s in res:
        assert s[0] == sref[0] and np.abs(p - ref).max() < 1e-8, (rank, np.abs(p - ref).max())
    for r in range(1, world):
        assert (res[0][1] == res[r][1]).all()


def test_emulated_ranks_replay_reproduces_lockstep():
    """tools/emulate_ranks.py (bench.py --emulate-rank): three ranks in lock step inside one process over the host-callback
    transport, then every rank ALONE through the device-callback transport replaying the recorded collectives: same trajectory to
    the bit, and the single-rank answer to rounding."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import emulate_ranks as E
    wl = dict(F=6, N=700, M=480)
    F, raws, poses, alts, grs = _survey()
    ref, ref_stats, _ = _reference(F, raws, poses, alts, grs)
    log, traj, stats, data = E.record(wl, 91, 3)                    # seed 91 = _survey()'s survey
    assert stats[0] == ref_stats[0] and np.abs(traj - ref).max() < 1e-9
    for r in range(3):
        ms, comm, brk, out = E.replay_rank(wl, 3, r, log[r], data, steps=1, warmup=1, profile=True)
        assert (out == traj).all()
        assert comm["allreduce_bytes"] > 0 and comm["allgather_bytes"] > 0 and brk["pg"] > 0
