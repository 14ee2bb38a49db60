"""world_size-2 gloo test (CPU) of the multi-rank plumbing: block sharding of frames, pairs to the owner of their target
frame, and the host-callback transport that diasss_amd/pipeline.py:make_comm hands to the library (all-reduce of doubles,
all-gather of bytes -- the two collectives dsss_comm.hip issues).  The device context is replaced by a recording stub (there
is no GPU here and no CPU fallback in the product); the numeric side of the partitioned pose-graph solve is pinned on the CPU
by tests/test_pg_symbolic_host.py, and end to end on the GPU by tests/test_gpu_multirank.py."""
import os
import sys
import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class StubCtx:
    """stands in for capi.Context: records calls; the collectives go through the callback exactly as libdsss.so drives it"""
    NB = 64

    def __init__(self, F):
        self.F = F; self.have = {}; self.matched = None; self.fn = None

    def comm_init_callback(self, rank, world, fn):
        self.rank, self.world, self.fn = rank, world, fn

    def extract_many(self, ids):
        for f in ids:
            self.have[int(f)] = np.full(self.NB, 10 + int(f), np.uint8)

    def features_allgather(self, F):                       # what dsss_features_allgather does: padded slices, own slice in place
        per = max(F * (r + 1) // self.world - F * r // self.world for r in range(self.world))
        buf = np.zeros((self.world, per * self.NB), np.uint8)
        f0, f1 = F * self.rank // self.world, F * (self.rank + 1) // self.world
        for f in range(f0, f1):                            # own frames only (a second step finds the others' records still there)
            buf[self.rank, (f - f0) * self.NB:(f - f0 + 1) * self.NB] = self.have[f]
        self.fn(1, buf)
        for r in range(self.world):
            g0, g1 = F * r // self.world, F * (r + 1) // self.world
            for f in range(g0, g1):
                self.have[f] = buf[r, (f - g0) * self.NB:(f - g0 + 1) * self.NB].copy()

    def match_pairs(self, src, tgt):
        self.matched = list(zip(src.tolist(), tgt.tolist()))

    def lc_solve_all(self):
        pass

    def posegraph_solve(self, F, total, want_rpy=False, pinned=False):
        # what the solve does with its partial sums: one all-reduce; here: one slot per target pose of every matched pair
        part = np.zeros(100 * F)
        for s, t in self.matched:
            part[100 * t + s] += 1.0
        self.fn(0, part)
        return part, None, np.zeros(4)


def _worker(rank, world, port, F, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from diasss_amd.pipeline import Pipeline, make_comm
    stub = StubCtx(F)
    make_comm(stub, dist, rank, world)
    pipe = Pipeline(F, rank=rank, world=world, dist=dist, ctx=stub)
    pipe.N = [100] * F; pipe.poses = [np.zeros((100, 6))] * F
    pipe.extract()
    pipe.match()
    summed, _ = pipe.optimize()
    q.put((rank, sorted(stub.have.keys()), [int(stub.have[f][0]) for f in sorted(stub.have)], stub.matched, np.nonzero(summed)[0].tolist(), float(summed.max())))
    dist.destroy_process_group()


@pytest.mark.parametrize("F", [5, 8])
def test_two_rank_sharding_gloo(F):
    from diasss_amd.pipeline import all_pairs, frame_owner, shard_frames, shard_pairs
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
    own = frame_owner(F, world)
    seen = []
    for rank, frames, tags, matched, nz, mx in res:
        assert frames == list(range(F))                       # every rank holds every frame after the all-gather
        assert tags == [10 + f for f in range(F)]             # ... with the record its owner produced
        assert all(own[t] == rank for _, t in matched)        # pairs go to the owner of the target frame
        assert nz == all_b and mx == 1.0                      # the all-reduce saw every pair exactly once, on every rank
        seen += matched
    assert sorted(seen) == sorted(zip(src.tolist(), tgt.tolist()))     # each pair matched exactly once
    assert shard_frames(F, 0, 2) + shard_frames(F, 1, 2) == list(range(F))        # contiguous blocks, in rank order
    s0, t0 = shard_pairs(src, tgt, 0, 2, F); s1, t1 = shard_pairs(src, tgt, 1, 2, F)
    assert len(s0) + len(s1) == len(src)


def test_frame_owner_matches_library():
    """the Python block rule is the library's (dsss_comm_frame_owner); checked without a device"""
    from diasss_amd import capi
    from diasss_amd.pipeline import frame_owner
    L = capi.lib()
    for F, world in ((5, 1), (200, 8), (7, 3)):
        own = frame_owner(F, world)
        # a null context means one rank: only the world == 1 rule can be asked from the library without a device
        if world == 1:
            assert [L.dsss_comm_frame_owner(None, F, f) for f in range(F)] == own.tolist()
        assert (np.diff(own) >= 0).all() and own[0] == 0 and own[-1] == world - 1


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` started plainly (no torch.distributed.run, WORLD_SIZE unset) starts one child per rank before
    anything touches a GPU, and rank 0 prints ONE JSON line with n_gpus 2.  --dry-run: gloo ranks on the recording stub above
    (this container has no GPU); the line says so and carries no value."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["DSSS_BENCH_CTX"] = "tests.test_distributed_gloo:StubCtx"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--workload", "smoke", "--dry-run"],
                         env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == 2 and rec["warmup"] == 1 and rec["dry_run"] is True and rec["value"] is None
    # a rank that fails takes the launcher down with a non-zero status instead of leaving the others in a collective
    env["DSSS_BENCH_CTX"] = "tests.test_distributed_gloo:NoSuchStub"
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "smoke", "--dry-run"], env=env, capture_output=True, text=True, timeout=300)
    assert bad.returncode != 0
