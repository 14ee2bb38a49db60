"""GPU: the real multi-rank pipeline (frames / pairs sharded over 2 ranks, feature all-gather, LC-edge all-gather,
replicated pose graph) must reproduce the single-rank trajectory.  Both ranks share the one GPU of the test box and
talk over gloo (RCCL refuses two ranks on one device); the collectives carry the same packed records as under RCCL."""
import os
import sys
import numpy as np
import pytest
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _survey():
    from diasss_amd.synth import Survey
    F, N, M = 4, 700, 480
    sv = Survey(F, N, M, seed=91)
    raws = [sv.frame(f).numpy() for f in range(F)]
    poses = [sv.inputs(f)[0] for f in range(F)]; alts = [sv.inputs(f)[1] for f in range(F)]; grs = [sv.inputs(f)[2] for f in range(F)]
    return F, raws, poses, alts, grs


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from diasss_amd.pipeline import Pipeline, shard_frames
    F, raws, poses, alts, grs = _survey()
    mine = set(shard_frames(F, rank, world))
    raws = [r if f in mine else None for f, r in enumerate(raws)]
    pipe = Pipeline(F, device=0, rank=rank, world=world, dist=dist)
    out, stats = pipe.run(raws, poses, alts, grs)
    q.put((rank, out, stats, pipe.n_edges))
    pipe.close()
    dist.destroy_process_group()


def test_two_ranks_equal_single_rank():
    from diasss_amd.pipeline import Pipeline
    F, raws, poses, alts, grs = _survey()
    pipe = Pipeline(F, device=0)
    ref, ref_stats = pipe.run(raws, poses, alts, grs)
    n_edges_ref = len(pipe.ctx.posegraph_select(F))
    pipe.close()
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29600 + os.getpid() % 1000
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs: p.start()
    res = sorted([q.get(timeout=300) for _ in range(world)], key=lambda t: t[0])
    for p in procs: p.join(timeout=60)
    assert n_edges_ref > 10
    for rank, out, stats, n_edges in res:
        assert n_edges == n_edges_ref
        assert stats[0] == ref_stats[0]
        assert np.abs(out - ref).max() < 1e-9
    assert (res[0][1] == res[1][1]).all()          # replicated solve is bit-identical across ranks


def _rccl_worker(port, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    from diasss_amd.pipeline import Pipeline
    F, raws, poses, alts, grs = _survey()
    pipe = Pipeline(F, device=0, rank=0, world=1, dist=dist, force_collectives=True)
    out, stats = pipe.run(raws, poses, alts, grs)
    q.put((out, stats, pipe.n_edges))
    pipe.close()
    dist.destroy_process_group()


def test_rccl_single_rank_collectives_path():
    """backend "nccl" (RCCL) with one rank on the one GPU: the feature records and LC edges go through
    all_gather_into_tensor on DEVICE tensors (pack / unpack with device pointers), result equal to the plain path"""
    from diasss_amd.pipeline import Pipeline
    F, raws, poses, alts, grs = _survey()
    pipe = Pipeline(F, device=0)
    ref, ref_stats = pipe.run(raws, poses, alts, grs)
    pipe.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    pr = ctx.Process(target=_rccl_worker, args=(29700 + os.getpid() % 1000, q))
    pr.start()
    out, stats, n_edges = q.get(timeout=300)
    pr.join(timeout=60)
    assert n_edges > 10 and stats[0] == ref_stats[0]
    assert np.abs(out - ref).max() < 1e-9
