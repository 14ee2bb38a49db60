"""world_size-2 gloo test (CPU) of the multi-rank plumbing in diasss_amd/pipeline.py: frame / pair sharding, the
all-gather of packed feature records and of the selected LC edges.  The device context is replaced by a recording
stub (there is no GPU here and no CPU fallback in the product); what is checked is that every rank ends up with every
frame's record, that each pair is matched exactly once by the owner of its target frame, and that the merged edge
list equals the single-rank list in the reference's order (ascending target pose id)."""
import os
import sys
import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class StubCtx:
    """stands in for capi.Context: records calls, (un)packs a fake per-frame record"""
    NB = 64

    def __init__(self, F):
        self.F = F; self.have = {}; self.matched = None

    def pack_bytes(self):
        return self.NB

    def extract_many(self, ids):
        for f in ids:
            self.have[int(f)] = np.full(self.NB, 10 + int(f), np.uint8)

    def features_pack(self, f, buf):
        buf.copy_(torch.from_numpy(self.have[int(f)]))

    def features_unpack(self, f, buf):
        self.have[int(f)] = buf.cpu().numpy().copy()

    def match_pairs(self, src, tgt):
        self.matched = list(zip(src.tolist(), tgt.tolist()))

    def lc_solve_all(self):
        pass

    def posegraph_select(self, F, cap=0):
        from diasss_amd import capi
        e = np.zeros(len(self.matched), capi.LCEDGE_DTYPE)
        for k, (s, t) in enumerate(self.matched):          # one fake edge per pair, target pose id = 100*t + s
            e["a"][k] = 100 * s; e["b"][k] = 100 * t + s; e["var"][k] = 1.0
        return e[np.argsort(e["b"], kind="stable")]

    def posegraph_solve_edges(self, dr, edges):
        return edges, np.zeros(4)


def _worker(rank, world, port, F, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from diasss_amd.pipeline import Pipeline
    pipe = Pipeline(F, rank=rank, world=world, dist=dist, ctx=StubCtx(F))
    pipe.N = [100] * F; pipe.poses = [np.zeros((100, 6))] * F
    pipe.extract()
    pipe.match()
    edges, _ = pipe.optimize()
    q.put((rank, sorted(pipe.ctx.have.keys()), [int(v[0]) for v in (pipe.ctx.have[f] for f in sorted(pipe.ctx.have))],
           pipe.ctx.matched, edges["b"].tolist()))
    dist.destroy_process_group()


@pytest.mark.parametrize("F", [5, 8])
def test_two_rank_sharding_gloo(F):
    from diasss_amd.pipeline import all_pairs, shard_frames, shard_pairs
    world = 2
    port = 29500 + (os.getpid() % 2000) + F
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, F, q)) for r in range(world)]
    for p in procs: p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs: p.join(timeout=60)
    res.sort()
    src, tgt = all_pairs(F)
    all_b = sorted(100 * int(t) + int(s) for s, t in zip(src, tgt))
    seen = []
    for rank, frames, tags, matched, eb in res:
        assert frames == list(range(F))                       # every rank holds every frame after the all-gather
        assert tags == [10 + f for f in range(F)]             # ... with the record its owner produced
        assert all(t % world == rank for _, t in matched)     # pairs go to the owner of the target frame
        assert eb == all_b                                    # merged LC edges: complete and in reference order
        seen += matched
    assert sorted(seen) == sorted(zip(src.tolist(), tgt.tolist()))     # each pair matched exactly once
    assert shard_frames(F, 0, 2) + shard_frames(F, 1, 2) != [] and len(set(shard_frames(F, 0, 2)) & set(shard_frames(F, 1, 2))) == 0
    s0, t0 = shard_pairs(src, tgt, 0, 2); s1, t1 = shard_pairs(src, tgt, 1, 2)
    assert len(s0) + len(s1) == len(src)
