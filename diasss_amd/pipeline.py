"""Host-side driver of the hot path over the C ABI (Python flavour of src/diasss2.cpp:83-101).

    for every frame:   Frame(...)                      -> dsss_frame_set + dsss_extract
    for every i < j:   FEAmatcher::RobustMatching      -> dsss_match_pairs (batched, all pairs in one call)
    Optimizer::TrajOptimizationAll                     -> dsss_lc_solve_all + dsss_posegraph_solve

One process per GPU.  With world_size > 1 (torch.distributed, backend nccl == RCCL on ROCm, or gloo on CPU for the
sharding logic tests) the work is sharded as:
    extraction   frames round-robin over ranks, then ONE all-gather of the packed per-frame feature records;
    matching+LC  every pair (s, t) goes to the rank that owns target frame t, so the "last pair wins" loop-closure
                 selection (optimizer.cpp:203-231) stays rank-local; then ONE all-gather of the selected LC edges;
    pose graph   replicated batch LM (identical on every rank).
"""
import numpy as np

from . import capi


def all_pairs(F):
    """the (i, j), i < j loop of diasss2.cpp:88-89, in that order"""
    i, j = np.triu_indices(F, 1)
    return i.astype(np.int32), j.astype(np.int32)


def shard_frames(F, rank, world):
    return [f for f in range(F) if f % world == rank]


def shard_pairs(src, tgt, rank, world):
    keep = (tgt % world) == rank
    return src[keep], tgt[keep]


def merge_edges(edge_lists):
    """concatenate per-rank LC edge arrays and restore the reference's order (ascending target pose id)"""
    edges = np.concatenate([e for e in edge_lists if len(e)]) if any(len(e) for e in edge_lists) else np.zeros(0, capi.LCEDGE_DTYPE)
    if len(edges):
        edges = edges[np.argsort(edges["b"], kind="stable")]
    return edges


class Pipeline:
    def __init__(self, F, device=0, rank=0, world=1, dist=None, min_overlap=None, ctx=None, force_collectives=False):
        self.F, self.rank, self.world, self.dist = F, rank, world, dist
        self.collectives = world > 1 or (force_collectives and dist is not None)   # the 1-rank RCCL test drives the same code path
        self.ctx = ctx if ctx is not None else capi.Context(max_frames=F, device=device)   # ctx injection: sharding tests
        self.min_overlap = min_overlap      # None: dense all-pairs (BASELINE configs); 0.4 reproduces diasss2.cpp:28,93

    def close(self):
        self.ctx.close()

    # ---- stage 1: frames
    def set_frames(self, raws, poses, alts, grs):
        """raws[f] may be None for frames this rank does not extract"""
        self.N = [p.shape[0] for p in poses]
        self.M = [2 * len(g) for g in grs]
        self.poses = poses
        if hasattr(self.ctx, "frames_set"):
            self.ctx.frames_set(list(range(self.F)), raws, self.N, self.M, poses, alts, grs)
        else:
            for f in range(self.F):
                self.ctx.frame_set(f, raws[f], self.N[f], self.M[f], poses[f], alts[f], grs[f])

    def extract(self):
        mine = shard_frames(self.F, self.rank, self.world)
        self.ctx.extract_many(mine)
        if self.collectives:
            self._allgather_features(mine)

    def _allgather_features(self, mine):
        import torch
        dist = self.dist
        nb = self.ctx.pack_bytes()
        per = (self.F + self.world - 1) // self.world
        dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
        send = torch.zeros((per, nb), dtype=torch.uint8, device=dev)
        for k, f in enumerate(mine):
            self.ctx.features_pack(f, send[k])
        recv = torch.empty((self.world * per, nb), dtype=torch.uint8, device=dev)     # concatenated along dim 0
        dist.all_gather_into_tensor(recv, send)
        if dev == "cuda":
            torch.cuda.synchronize()      # the library reads `recv` on its own HIP stream: order it after the RCCL collective
        for r in range(self.world):
            if r == self.rank and self.world > 1:
                continue                   # (a 1-rank group unpacks its own records: exercises the device-pointer path)
            for k, f in enumerate(shard_frames(self.F, r, self.world)):
                self.ctx.features_unpack(f, recv[r * per + k])

    # ---- stage 2: matching + loop-closure measurements
    def match(self):
        src, tgt = all_pairs(self.F)
        if self.min_overlap is not None:
            keep = np.array([self.ctx.overlap(int(i), int(j)) > self.min_overlap for i, j in zip(src, tgt)], bool)
            src, tgt = src[keep], tgt[keep]
        self.src, self.tgt = shard_pairs(src, tgt, self.rank, self.world)
        self.ctx.match_pairs(self.src, self.tgt)
        self.ctx.lc_solve_all()

    # ---- stage 3: pose graph
    def optimize(self):
        total = int(sum(self.N))
        if not self.collectives:
            # est_poses into a page-locked buffer of the context (valid until the next solve); the rpy rows are only for SaveTrajactoryAll
            poses, _, stats = self.ctx.posegraph_solve(self.F, total, want_rpy=False, pinned=True)
            self.n_edges = None
            return poses, stats
        edges = self.ctx.posegraph_select(self.F, cap=max(total, 1))
        edges = self._allgather_edges(edges)
        self.n_edges = len(edges)
        dr = np.concatenate(self.poses)
        return self.ctx.posegraph_solve_edges(dr, edges)

    def _allgather_edges(self, edges):
        import torch
        dist = self.dist
        dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
        cnt = torch.tensor([len(edges)], dtype=torch.int64, device=dev)
        cnts = [torch.zeros_like(cnt) for _ in range(self.world)]
        dist.all_gather(cnts, cnt)
        cnts = [int(c.item()) for c in cnts]
        mx = max(max(cnts), 1)
        isz = capi.LCEDGE_DTYPE.itemsize
        buf = np.zeros(mx * isz, np.uint8)
        buf[:len(edges) * isz] = edges.view(np.uint8).reshape(-1)
        send = torch.from_numpy(buf).to(dev)
        recv = torch.empty(self.world * mx * isz, dtype=torch.uint8, device=dev)
        dist.all_gather_into_tensor(recv, send)
        if dev == "cuda":
            torch.cuda.synchronize()
        recv = recv.cpu().numpy().reshape(self.world, mx * isz)
        return merge_edges([recv[r, :cnts[r] * isz].copy().view(capi.LCEDGE_DTYPE) for r in range(self.world)])

    def run(self, raws, poses, alts, grs):
        self.set_frames(raws, poses, alts, grs)
        self.extract()
        self.match()
        return self.optimize()
