"""Host-side driver of the hot path over the C ABI (Python flavour of src/diasss2.cpp:83-101).

    for every frame:   Frame(...)                      -> dsss_frame_set + dsss_extract
    for every i < j:   FEAmatcher::RobustMatching      -> dsss_match_pairs (batched, all pairs in one call)
    Optimizer::TrajOptimizationAll                     -> dsss_lc_solve_all + dsss_posegraph_solve

One process per GPU.  With world_size > 1 the ranks form a communicator INSIDE the library (dsss_comm_init: RCCL over xGMI;
torch.distributed only carries the 128-byte RCCL id to the other ranks -- or, with the gloo backend on a one-GPU test box,
serves as the host-callback transport) and the work is sharded as:
    extraction   contiguous blocks of frames per rank, then ONE all-gather of the packed per-frame feature records;
    matching+LC  every pair (s, t) goes to the rank that owns target frame t, so the "last pair wins" loop-closure
                 selection (optimizer.cpp:203-231) stays rank-local;
    pose graph   every rank eliminates the poses of its own frames; ONE all-reduce per LM trial sums the reduced Hessian on
                 the interface poses (interface blocks | Schur complements | gradient), then a small replicated interface
                 solve and the local back-substitution (dsss_pg.hip).
"""
import numpy as np

from . import capi


def all_pairs(F):
    """the (i, j), i < j loop of diasss2.cpp:88-89, in that order"""
    i, j = np.triu_indices(F, 1)
    return i.astype(np.int32), j.astype(np.int32)


def shard_frames(F, rank, world):
    """contiguous block of frames of a rank: [F rank / world, F (rank + 1) / world) -- dsss_comm_frame_owner"""
    return list(range(F * rank // world, F * (rank + 1) // world))


def frame_owner(F, world):
    own = np.zeros(F, np.int64)
    for r in range(world):
        own[F * r // world:F * (r + 1) // world] = r
    return own


def shard_pairs(src, tgt, rank, world, F=None):
    """pairs whose TARGET frame this rank owns"""
    F = int(max(src.max(initial=0), tgt.max(initial=0)) + 1) if F is None else F
    keep = frame_owner(F, world)[tgt] == rank
    return src[keep], tgt[keep]


def merge_edges(edge_lists):
    """concatenate per-rank LC edge arrays and restore the reference's order (ascending target pose id)"""
    edges = np.concatenate([e for e in edge_lists if len(e)]) if any(len(e) for e in edge_lists) else np.zeros(0, capi.LCEDGE_DTYPE)
    if len(edges):
        edges = edges[np.argsort(edges["b"], kind="stable")]
    return edges


def make_comm(ctx, dist, rank, world):
    """join the library's communicator: RCCL when torch.distributed runs on nccl, the host callback over gloo otherwise"""
    import torch
    if dist.get_backend() == "nccl":
        uid = torch.from_numpy(ctx.comm_unique_id() if rank == 0 else np.zeros(128, np.uint8)).cuda()
        dist.broadcast(uid, 0)
        torch.cuda.synchronize()
        ctx.comm_init_rccl(uid.cpu().numpy(), rank, world)
    else:
        def fn(op, arr):
            t = torch.from_numpy(arr)
            if op == 0:
                dist.all_reduce(t)
            else:
                mine = t[rank].clone()
                dist.all_gather_into_tensor(t.view(-1), mine)
        ctx.comm_init_callback(rank, world, fn)


class Pipeline:
    def __init__(self, F, device=0, rank=0, world=1, dist=None, min_overlap=None, ctx=None, force_collectives=False, nfeatures=None):
        self.F, self.rank, self.world, self.dist = F, rank, world, dist
        self.ctx = ctx if ctx is not None else capi.Context(max_frames=F, device=device)   # ctx injection: sharding tests
        if nfeatures is not None:            # frame.cpp:180 hard-codes ORBextractor(2000, ...); BASELINE config 5 asks for 8000
            _, op, _, _ = self.ctx.default_params()
            op.nfeatures = int(nfeatures)
            self.ctx.set_params(orb=op)
        self.min_overlap = min_overlap      # None: dense all-pairs (BASELINE configs); 0.4 reproduces diasss2.cpp:28,93
        if (world > 1 or force_collectives) and dist is not None and ctx is None:
            make_comm(self.ctx, dist, rank, world)       # a 1-rank RCCL communicator drives the same code path

    def close(self):
        self.ctx.close()

    # ---- stage 1: frames
    def prepare(self, raws, poses, alts, grs):
        """the argument arrays of dsss_frames_set for one survey, built and validated ONCE (capi.FramesArgs: what a C++ caller of the C ABI
        holds anyway); run(prepared) / set_frames(prepared) then go straight to the C call.  raws[f] may be None for frames this rank
        does not extract."""
        N = [p.shape[0] for p in poses]; M = [2 * len(g) for g in grs]
        args = self.ctx.frames_args(list(range(self.F)), raws, N, M, poses, alts, grs) if hasattr(self.ctx, "frames_args") else None
        return dict(args=args, N=N, M=M, raws=raws, poses=poses, alts=alts, grs=grs)

    def set_frames(self, raws, poses=None, alts=None, grs=None):
        h = raws if isinstance(raws, dict) else self.prepare(raws, poses, alts, grs)
        self.N, self.M, self.poses = h["N"], h["M"], h["poses"]
        if h["args"] is not None:
            self.ctx.frames_set(h["args"])
        else:                                              # (a recording stub of the CPU tests)
            for f in range(self.F):
                self.ctx.frame_set(f, h["raws"][f], self.N[f], self.M[f], h["poses"][f], h["alts"][f], h["grs"][f])

    def extract(self):
        self.ctx.extract_many(shard_frames(self.F, self.rank, self.world))
        if self.world > 1:
            self.ctx.features_allgather(self.F)      # one all-gather of the packed feature records, inside the library

    # ---- stage 2: matching + loop-closure measurements
    def match(self):
        if self.min_overlap is not None or getattr(self, "_pairs", None) is None:
            src, tgt = all_pairs(self.F)
            if self.min_overlap is not None:
                keep = np.array([self.ctx.overlap(int(i), int(j)) > self.min_overlap for i, j in zip(src, tgt)], bool)
                src, tgt = src[keep], tgt[keep]
            self._pairs = shard_pairs(src, tgt, self.rank, self.world, self.F)     # dense all-pairs: the list depends on F and the rank only
        self.src, self.tgt = self._pairs
        self.ctx.match_pairs(self.src, self.tgt)
        self.ctx.lc_solve_all()

    # ---- stage 3: pose graph (sharded inside the library when the context joined a communicator)
    def optimize(self):
        total = int(sum(self.N))
        # est_poses into a page-locked buffer of the context (valid until the next solve); the rpy rows are only for SaveTrajactoryAll
        poses, _, stats = self.ctx.posegraph_solve(self.F, total, want_rpy=False, pinned=True)
        self.n_edges = None
        return poses, stats

    def run(self, raws, poses=None, alts=None, grs=None):
        """one survey through the whole path; `raws` may be the handle of prepare()"""
        self.set_frames(raws, poses, alts, grs)
        self.extract()
        self.match()
        return self.optimize()
