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


def _reference(F, raws, poses, alts, grs):
    from diasss_amd.pipeline import Pipeline
    pipe = Pipeline(F, device=0)
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


def test_rccl_one_rank_communicator():
    F, raws, poses, alts, grs = _survey()
    ref, ref_stats, n_edges = _reference(F, raws, poses, alts, grs)
    (rank, out, stats, repro, cs), = _run(1, "nccl", 29700 + os.getpid() % 1000)
    assert stats[0] == ref_stats[0] and np.abs(out - ref).max() < 1e-9 and repro
    assert cs[1] == 1 and cs[3] > 0
