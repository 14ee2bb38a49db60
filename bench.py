#!/usr/bin/env python3
"""bench.py -- end-to-end sonar frames/s of the MI355X-native diasss hot path (BASELINE.json metric).

A "step" is one pass of the whole hot path over one synthetic survey: Frame construction (normalise, mask,
geo box, ORB extraction) for every frame, dense all-pairs FEAmatcher::RobustMatching, sonar reprojection,
the batched mini-LM loop-closure solve and the pose-graph Levenberg-Marquardt solve.  The raw waterfalls are
resident in HBM before the timed region starts.

    python bench.py --gpus 1 --steps K --warmup W                         (default workload: BASELINE config 3)
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...   (one rank per GPU, RCCL)

Prints ONE JSON line on rank 0 (contract in the task statement) with the extra `roofline` and `cpu_baseline`
objects.  The CPU baseline is the oracle (oracle/liboracle.so, a single-threaded C restatement of the reference:
the reference itself needs OpenCV/GTSAM and cannot be built here) timed on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # BASELINE.json configs[1..2]; "smoke" is for quick checks only
    "C2": dict(F=50, N=1000, M=512, name="50 synthetic 1000x512 frames, dense all-pairs"),
    "C3": dict(F=200, N=2000, M=1024, name="200 synthetic 2000x1024 frames, dense all-pairs"),
    # BASELINE.json configs[4] on ONE GPU (the config names 8; the whole survey fits one 288 GB device): --workload C5
    "C5": dict(F=1000, N=4000, M=2048, nfeatures=8000, name="1000 synthetic 4000x2048 frames, 8k kp/frame, dense all-pairs, long-trajectory pose graph"),
    "smoke": dict(F=6, N=700, M=480, name="6 synthetic 700x480 frames (not a BASELINE config)"),
}
HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy ceiling)


def cpu_baseline(sv, wl, n_frames, threads=1):
    """oracle on the first n_frames frames of the same survey: extraction, all pairs, LC, pose graph.
    threads = 1: the reference's own execution model (it is single-threaded).  threads > 1: frames and frame pairs spread
    over a thread pool (the C calls release the GIL); the pose-graph solve stays on one core."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import binding as O
    N, M = wl["N"], wl["M"]
    raws = [sv.frame(f).cpu().numpy() for f in range(n_frames)]
    ins = [sv.inputs(f) for f in range(n_frames)]
    O.lib()

    po = None
    if wl.get("nfeatures"):
        po = O.orb_params(); po.nfeatures = wl["nfeatures"]

    def extract(f):
        kps, desc, _, _ = O.detect_feature(raws[f], None, po) if po is not None else O.detect_feature(raws[f])
        return kps, desc
    pool = ThreadPoolExecutor(max_workers=threads) if threads > 1 else None
    mp = (lambda fn, it: list(pool.map(fn, it))) if pool else (lambda fn, it: [fn(v) for v in it])
    t1 = time.time()
    feats = mp(extract, range(n_frames))
    t_extract = time.time() - t1
    fr = []
    for f in range(n_frames):                                   # python loop over keypoints: not part of the measurement
        pose, alt, gr = ins[f]
        fr.append(dict(pose=pose, alt=alt, gr=gr, kps=feats[f][0], desc=feats[f][1], geo=O.geo_at_kps(pose, gr, M, feats[f][0]), bb=O.geo_bbox(pose, gr, M)))
    pairs = [(i, j) for i in range(n_frames) for j in range(i + 1, n_frames)]

    def do_pair(ij):
        i, j = ij
        a, b = fr[i], fr[j]
        rows = O.robust_matching(i, j, N, N, a["kps"], a["desc"], a["geo"], a["bb"], b["kps"], b["desc"], b["geo"], b["bb"])
        kp7 = O.get_kps_pairs(rows, j, a["alt"], a["gr"], b["alt"], b["gr"])
        return kp7, O.lc_solve(kp7, a["pose"], a["alt"], a["gr"], M, b["pose"], b["alt"], b["gr"], M)
    t2 = time.time()
    res = mp(do_pair, pairs)
    pair_off = [0]
    for kp7, _ in res:
        pair_off.append(pair_off[-1] + len(kp7))
    kp7_all = np.concatenate([r[0] for r in res]) if res else np.zeros((0, 7))
    lcs_all = np.concatenate([r[1] for r in res]) if res else np.zeros(0, O.LC_DTYPE)
    edges = O.pg_select_lc([N] * n_frames, [p[0] for p in pairs], [p[1] for p in pairs], pair_off, kp7_all, lcs_all)
    dr = np.concatenate([f["pose"] for f in fr])
    t3 = time.time()
    o_poses, _ = O.pg_solve(dr, edges, solver="sparse")          # the oracle's LM with its reduced system through a sparse LU (round 6: the envelope Cholesky it
    t_pg = time.time() - t3                                      # used before took 4.5 of this leg's 4.8 s -- GTSAM's sparse solvers would not)
    t_rest = time.time() - t2
    if pool:
        pool.shutdown()
    t_cpu = t_extract + t_rest
    return dict(value=n_frames / t_cpu, unit="frames/s", cores=threads, kind="port",
                sample="oracle (C restatement of the reference, -O3, %d thread%s) on %d of %d frames of %dx%d: extraction %.1fs, "
                       "all %d pairs + LC + pose graph %.1fs (pose graph %.1fs%s)" % (threads, "" if threads == 1 else "s", n_frames, wl["F"], N, M, t_extract, len(pairs), t_rest, t_pg,
                                                                                  ", on ONE core whatever the thread count: the threads speed up extraction and matching only" if threads > 1 else "")), o_poses, len(edges)


def sample_parity(sv, n_frames, o_poses, o_edges, device, nfeatures=None):
    """the HIP path on the CPU sample's frames against the oracle's trajectory (pose RMSE vs the oracle's batch LM)"""
    from diasss_amd.pipeline import Pipeline
    pipe = Pipeline(n_frames, device=device, nfeatures=nfeatures)
    raws = [sv.frame(f) for f in range(n_frames)]
    ins = [sv.inputs(f) for f in range(n_frames)]
    poses, stats = pipe.run(raws, [i[0] for i in ins], [i[1] for i in ins], [i[2] for i in ins])
    n_edges = len(pipe.ctx.posegraph_select(n_frames))
    pipe.close()
    dt = poses[:, 9:] - o_poses[:, 9:]
    Rg = poses[:, :9].reshape(-1, 3, 3); Ro = o_poses[:, :9].reshape(-1, 3, 3)
    ang = np.sqrt(((Rg - Ro) ** 2).sum((1, 2)) / 2.0)          # rotation angle of Rg^T Ro for small angles: |Rg - Ro|_F / sqrt(2)
    return {"frames": n_frames, "poses": int(len(poses)), "lc_edges_gpu": int(n_edges), "lc_edges_oracle": int(o_edges),
            "trans_rmse_m": float(np.sqrt((dt ** 2).sum(1).mean())), "rot_rmse_rad": float(np.sqrt((ang ** 2).mean())),
            "max_abs_pose_entry": float(np.abs(poses - o_poses).max())}


def launch_ranks(n, argv):
    """one child process per GPU, as `python -m torch.distributed.run --nnodes=1 --nproc-per-node n` would start them; this
    process never initialises the GPU.  A child that fails takes the others down (they would wait in a collective for ever)."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")        # dmabuf IPC: RCCL between processes needs it on this driver
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env))
    rc = 0
    live = list(procs)
    while live:
        for p in list(live):
            try:
                p.wait(timeout=0.2)
            except subprocess.TimeoutExpired:
                continue
            live.remove(p)
            if p.returncode != 0 and rc == 0:
                rc = p.returncode
                for q in live:
                    q.terminate()                               # exactly the children started above
    return rc


def dry_run(args, rank, world):
    """the multi-rank plumbing of this file without a device: rendezvous over gloo, the sharded Pipeline on a context stub, the
    barrier + max-over-ranks reduction of the step time, ONE line from rank 0.  No frames, no kernels: value is null."""
    import importlib
    import torch
    import torch.distributed as dist
    from diasss_amd.pipeline import Pipeline, make_comm
    mod, cls = os.environ["DSSS_BENCH_CTX"].split(":")
    wl = WORKLOADS[args.workload]
    F = wl["F"]
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
    stub = getattr(importlib.import_module(mod), cls)(F)
    if world > 1:
        make_comm(stub, dist, rank, world)
    pipe = Pipeline(F, rank=rank, world=world, dist=dist if world > 1 else None, ctx=stub)
    pipe.N = [wl["N"]] * F; pipe.poses = [None] * F
    t0 = time.perf_counter()
    for _ in range(args.warmup + args.steps):
        pipe.extract(); pipe.match(); pipe.optimize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64); dist.all_reduce(t, op=dist.ReduceOp.MAX); dt = float(t.item())
    if rank == 0:
        print(json.dumps({"metric": "sonar frames/sec end-to-end (extract+match+LM solve)", "value": None, "unit": "frames/s", "n_gpus": world, "steps": args.steps,
                          "warmup": args.warmup, "ms_per_step": None, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dry_run": True,
                          "config": {"workload": wl["name"], "frames": F}}))
    if world > 1:
        dist.destroy_process_group()


def emulate_rank(args):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import emulate_ranks as E
    which, world = args.emulate_rank.split("/")
    world = int(world)
    wl = WORKLOADS[args.workload]
    ranks = None if which == "all" else [int(which)]
    rows, stats = E.emulate(wl, 20240601 + ["C2", "C3", "smoke", "C5"].index(args.workload), world, ranks, steps=args.steps, warmup=args.warmup, profile=True)
    worst = max(r["ms_per_step"] + r["wire_ms_model"] for r in rows)
    print(json.dumps({"metric": "per-rank step time of a %d-rank job, emulated on one GPU by replaying recorded collectives" % world, "unit": "ms", "world": world,
                      "workload": wl["name"], "ranks": rows, "lm_iterations": int(stats[0]),
                      "implied_step_ms": worst if ranks is None else None, "implied_frames_per_s": wl["F"] / worst * 1e3 if ranks is None else None,
                      "wire_model": "ring collectives bound by one xGMI link at %.0f GB/s: all-reduce 2 (W-1)/W x bytes, all-gather (W-1)/W x bytes" % E.XGMI_LINK_GBS}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default=os.environ.get("DSSS_WORKLOAD", "C3"), choices=sorted(WORKLOADS))
    ap.add_argument("--cpu-frames", type=int, default=24, help="frames in the CPU baseline sample (0 = skip); 24 frames of C3 are about 10 s of one core")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-side-legs", action="store_true", help="skip the legs that run extra passes after the timed steps (all-pairs matcher, SIFT descriptor): a kernel trace of the command then ends with the timed steps and the profiled pass")
    ap.add_argument("--pcie-steps", type=int, default=2, help="steps of the PCIe-inclusive leg (raw frames in page-locked host memory); 0 = skip")
    ap.add_argument("--jobs-in-flight", type=int, default=4, help="surveys overlapped in the extra throughput leg (1 = skip); the rate saturates at four on one MI355X")
    ap.add_argument("--pg-partitions", type=int, default=0, help="cut the pose graph into this many contiguous blocks of frames on the rank(s) present (dsss_set_pg_partitions): the layout of a "
                                                                 "P-rank job on fewer ranks, and on one GPU a nested dissection whose first levels run between frame blocks (0 = the library's default)")
    ap.add_argument("--dry-run", action="store_true", help="plumbing check without a GPU (CPU test-suite): gloo ranks, the context class named by DSSS_BENCH_CTX "
                                                           "(module:Class, a recording stub), no frames, no timing claim -- the line carries value null and dry_run true")
    ap.add_argument("--emulate-rank", default=None, metavar="r/W", help="time rank r (or `all`) of a W-rank job on ONE GPU: all W ranks run once in lock step inside "
                                                                       "this process and record the results of their collectives, then rank r runs alone and replays them "
                                                                       "(tools/emulate_ranks.py).  Prints its own JSON line, not the bench contract's")
    args = ap.parse_args()
    if args.emulate_rank:
        return emulate_rank(args)

    # `python bench.py --gpus N` started plainly (no torch.distributed.run): this process becomes the launcher.  It starts one
    # child per GPU BEFORE anything here has touched the GPU (nothing is imported that could), hands them the rendezvous through
    # the environment torch.distributed.run would have set, and exits with their status; rank 0 prints the JSON line.
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE is %d" % (args.gpus, world))
    if args.dry_run:
        return dry_run(args, rank, world)
    from diasss_amd import capi
    from diasss_amd.pipeline import Pipeline, shard_frames
    from diasss_amd.synth import Survey
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    # DSSS_BENCH_FORCE_COMM=1: drive the multi-rank code path (RCCL communicator inside the library, feature all-gather, edge exchange,
    # interface all-reduce) with ONE rank -- a rehearsal of the N > 1 launch on a one-GPU box; the pose graph is cut into 8 partitions
    force_comm = world == 1 and os.environ.get("DSSS_BENCH_FORCE_COMM") == "1"
    saved_stdout = None
    if world > 1 or force_comm:
        # RCCL prints a version banner on STDOUT when its first communicator comes up; stdout carries the one JSON line and nothing else,
        # so file descriptor 1 points at stderr until both communicators (torch's and the library's) exist
        sys.stdout.flush(); saved_stdout = os.dup(1); os.dup2(2, 1)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    wl = WORKLOADS[args.workload]
    F, N, M = wl["F"], wl["N"], wl["M"]
    big = F * N * M > (1 << 32)                        # C5: 65 GB of frames -- device-side noise, no host-resident or second-context legs by default
    if big:
        args.pcie_steps = 0 if args.pcie_steps == 2 else args.pcie_steps
        args.jobs_in_flight = 1 if args.jobs_in_flight == 4 else args.jobs_in_flight
        args.cpu_frames = min(args.cpu_frames, 6)
    sv = Survey(F, N, M, seed=20240601 + ["C2", "C3", "smoke", "C5"].index(args.workload), device="cuda:%d" % local_rank, noise_on_device=big)
    mine = shard_frames(F, rank, world)
    raws = [None] * F
    for f in mine:
        raws[f] = sv.frame(f)                       # float64 N x M, resident in HBM
    poses = [sv.inputs(f)[0] for f in range(F)]; alts = [sv.inputs(f)[1] for f in range(F)]; grs = [sv.inputs(f)[2] for f in range(F)]
    torch.cuda.synchronize()

    pipe = Pipeline(F, device=local_rank, rank=rank, world=world, dist=dist if (world > 1 or force_comm) else None, nfeatures=wl.get("nfeatures"), force_collectives=force_comm)
    if force_comm:
        pipe.ctx.set_pg_partitions(8)
    if args.pg_partitions > 0:
        pipe.ctx.set_pg_partitions(args.pg_partitions)
    if saved_stdout is not None:
        dist.barrier(); torch.cuda.synchronize()
        sys.stdout.flush(); os.dup2(saved_stdout, 1); os.close(saved_stdout)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        pipe.ctx.sync()

    # the argument arrays of dsss_frames_set (ids, sizes, four arrays of per-frame pointers) are built ONCE, before the clock: a C++ caller of
    # the C ABI holds them as plain arrays; building them from Python lists is the harness's cost, not the path's.  Every C-ABI call of a step
    # (dsss_frames_set: geometry staging + upload + start of the extraction, dsss_extract_many, dsss_match_pairs, dsss_lc_solve_all,
    # dsss_posegraph_solve) runs inside the timed region, on every step.
    survey = pipe.prepare(raws, poses, alts, grs)
    stats = None
    for _ in range(args.warmup):
        _, stats = pipe.run(survey)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        _, stats = pipe.run(survey)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # ---- per-kernel timing pass (HIP events on the library's stream), outside the timed region
    breakdown, roof, roof_groups, roof_all, work = None, None, None, None, None
    if not args.no_roofline:
        # three profiled steps, the MEDIAN time of every kernel scope: a scope's HIP events bracket its launches, so a stall of the launching
        # host thread in the middle of one (the shared hosts show one of 4 - 8 ms every few steps) reads as kernel time -- a single pass
        # once reported the extraction at 38.8 ms in a run whose whole step took 21.0; counts and work are the same in all three passes
        pipe.ctx.profile(True)
        passes = []
        for _ in range(3):
            pipe.ctx.profile_reset()
            pipe.run(survey)
            barrier()
            passes.append(pipe.ctx.profile_get())
        pipe.ctx.profile(False)
        prof = {k: (sorted(p[k][0] for p in passes)[1],) + tuple(passes[0][k][1:]) for k in passes[0]}
        # mini-LM work: 3e4 flops per LM iteration and problem (+ one linearisation for the marginal covariance)
        act = [p for p in range(len(pipe.src)) if pipe.ctx.pair_is_active(p)]
        lc_iters = 0
        for p in act:
            g = pipe.ctx.lc_get(p)
            lc_iters += int(g["iters"].sum()) + len(g)
        if "lc" in prof and prof["lc"][1] > 0:
            prof["lc"] = (prof["lc"][0], prof["lc"][1], 3e4 * lc_iters)
        breakdown = {k: round(v[0], 3) for k, v in prof.items() if v[1] > 0}
        trials = prof["pg_subtree"][1] // 2 if "pg_subtree" in prof and prof["pg_subtree"][1] > 0 else 1     # bins forward + backward once per LM trial
        roof, roof_groups, roof_all = roofline(prof, args.workload, wl, trials)
        work = {k: ("%.3g flop" if k in FLOP_SLOTS else "%.3g B") % v[2] for k, v in prof.items() if v[1] > 0 and v[2] > 0}
    # ---- side leg: the ALL-PAIRS matcher north_star describes (LDS-staged descriptor tiles, every keypoint of a against every keypoint of b).
    # The step runs the geo-grid matcher (identical outputs from 3 % of the gate evaluations); DSSS_MT_GRID=0 selects the all-pairs kernel,
    # which is timed here on the same features so that it keeps a measured number of its own.  Outside the timed region; the state of the
    # context is put back by matching once more under the default.
    match_allpairs = None
    if not args.no_roofline and not args.no_side_legs and world == 1 and not big:
        os.environ["DSSS_MT_GRID"] = "0"
        try:
            pipe.ctx.match_pairs(pipe.src, pipe.tgt); barrier()
            pipe.ctx.profile(True); pipe.ctx.profile_reset()
            pipe.ctx.match_pairs(pipe.src, pipe.tgt); barrier()
            pm = pipe.ctx.profile_get(); pipe.ctx.profile(False)
        finally:
            del os.environ["DSSS_MT_GRID"]
        pipe.ctx.match_pairs(pipe.src, pipe.tgt); pipe.ctx.lc_solve_all(); barrier()
        if pm.get("match", (0, 0, 0))[1] > 0 and pm["match"][0] > 0:
            nk = {f: pipe.ctx.features_get(f)[0].shape[0] for f in range(F)}
            actp = [p for p in range(len(pipe.src)) if pipe.ctx.pair_is_active(p)]
            evals = float(sum(2 * nk[int(pipe.src[p])] * nk[int(pipe.tgt[p])] for p in actp))          # both directions of every active pair
            alg_bytes = float(sum(2 * (nk[int(pipe.src[p])] + nk[int(pipe.tgt[p])]) * 56 for p in actp))  # (Na + Nb) (32 + 16 + 8) per direction
            sec = pm["match"][0] * 1e-3
            tr = pmc_traffic(args.workload).get("match_nn_kernel<false>")
            match_allpairs = {"kernel": "match_nn_kernel<false> (DSSS_MT_GRID=0)", "ms": pm["match"][0], "launches": pm["match"][1],
                              "evaluations": evals, "evaluations_per_s_G": evals / sec / 1e9, "peak_G": MATCH_PEAK_GEVALS, "frac": evals / sec / 1e9 / MATCH_PEAK_GEVALS,
                              # SURVEY K9's ceiling taken literally (256 CU x 64 lanes x 2.4 GHz / 30 operations): the kernel spends ~13.4 lane-instructions per
                              # evaluation, not 30, so it reads above 1 -- "30 operations per comparison" is an estimate, not a hardware bound; the honest figure is the
                              # issue utilisation of roofline_all_kernels (match_nn_kernel: ~0.6 of the measured full-rate issue slots)
                              "peak_G_survey_literal": 256 * 64 * 2.4e9 / 30.0 / 1e9, "frac_survey_literal": evals / sec / 1e9 / (256 * 64 * 2.4e9 / 30.0 / 1e9),
                              "algorithmic_bytes": alg_bytes, "algorithmic_GBs": alg_bytes / sec / 1e9,
                              "hbm_GBs": (tr[0] * pm["match"][1] / sec / 1e9) if tr else None,
                              "note": "every keypoint of a against every keypoint of b over the active pairs (gate in f64, then the 256-bit Hamming distance), descriptor tiles staged in LDS; "
                                      "bound by its integer issue slots, not by bytes: SURVEY 8(d) K9 prices it against lanes x clock / 30 operations"}
    # ---- side leg: N4 of SURVEY 8f -- the 128-float descriptor of the reference's SIFT call site and its L2 matcher (dsss_sift.hip, match_*<2>), on
    # the same survey: extraction with DSSS_DESC_SIFT128, matching with use_l2 = 2.  Outside the timed region; the context goes back to the
    # ORB / Hamming configuration (and to its features and matches) afterwards.
    sift_leg = None
    if not args.no_roofline and not args.no_side_legs and world == 1 and not big:
        from diasss_amd import capi as _capi
        mp_, op_, mt_, pg_ = pipe.ctx.default_params()
        if wl.get("nfeatures"):
            op_.nfeatures = wl["nfeatures"]
        op_s = type(op_).from_buffer_copy(op_); mt_s = type(mt_).from_buffer_copy(mt_)
        op_s.descriptor = _capi.DESC_SIFT128; mt_s.use_l2 = 2
        try:
            pipe.ctx.set_params(orb=op_s, match=mt_s)
            pipe.set_frames(survey); pipe.extract(); pipe.match(); barrier()          # first use: the 128-byte store
            pipe.ctx.profile(True); pipe.ctx.profile_reset()
            t_s = time.perf_counter()
            pipe.set_frames(survey); pipe.extract(); barrier()
            t_e = time.perf_counter()
            pipe.match(); barrier()
            t_m = time.perf_counter()
            ps = pipe.ctx.profile_get(); pipe.ctx.profile(False)
            rows_s, kp7_s = pipe.ctx.match_total()
            nkp_s = int(sum(pipe.ctx.features_get(f)[0].shape[0] for f in range(F)))
            k_ms = ps["sift"][0]
            sift_leg = {"kernel": "sift_desc_kernel", "ms": k_ms, "keypoints_pre_filter": int(F * op_s.nfeatures), "keypoints": nkp_s,
                        "algorithmic_bytes": ps["sift"][2], "algorithmic_GBs": ps["sift"][2] / (k_ms * 1e-3) / 1e9 if k_ms > 0 else None,
                        "us_per_keypoint": 1e3 * k_ms / max(1, F * op_s.nfeatures),
                        "extract_wall_ms_with_sift": 1e3 * (t_e - t_s), "match_l2_128_ms": ps["match"][0], "match_lc_wall_ms": 1e3 * (t_m - t_e),
                        "match_rows": int(rows_s), "lc_problems": int(kp7_s),
                        "note": "one workgroup per keypoint: 71 x 71 window of the level image, 13-tap 8.8 blur, 57 x 57 samples (gradient, fastAtan2, sqrt, table weight, "
                                "trilinear shares into a 2^-12 fixed-point histogram by LDS atomics), integer normalisation; rows bit-exact vs oracle/orc_sift.c "
                                "(tests/test_gpu_sift.py).  An optional mode (dsss_orb_params.descriptor = DSSS_DESC_SIFT128), not part of `value`"}
        finally:
            pipe.ctx.set_params(orb=op_, match=mt_)
            pipe.run(survey); barrier()
    # ---- the dependent chains of the factorisation: what the LM trials would cost if every level took only its longest chain of dependent
    # f64 operations (constants from the in-kernel stamps and tools/ubench/mfma_f64, see DESIGN.md "critical path")
    crit = None
    if not args.no_roofline:
        lv, ntr = pipe.ctx.posegraph_schedule()
        if len(lv):
            CLK = 2.29e9                                     # s_memtime ticks per second in these kernels (profiles/r04_front_kernel_stamps.txt)
            C_PIVOT, C_PROD, C_BWD = 570.0, 102.0, 312.0     # cycles: arithmetic of one 4-column pivot block; one dependent f64 matrix product (result -> operand); one block of the triangular back-substitution
            cyc = 0.0
            for items, w6, rows, _ in lv:
                nb = (int(w6) + 3) // 4
                cyc += nb * (C_PIVOT + 2 * C_PROD)           # panel Cholesky: pivot block -> two dependent products -> next pivot block
                if rows > 0:
                    cyc += nb * 2 * C_PROD                   # row solve: two dependent products per 4-column step
                cyc += nb * C_BWD                            # back-substitution of the panel
            crit = {"levels": int(len(lv)), "trials": int(ntr), "ms_per_trial": cyc / CLK * 1e3, "ms_per_step": cyc / CLK * 1e3 * ntr,
                    "note": "sum over the panel levels of the schedule of the longest DEPENDENT f64 chain of each level (4-column pivot blocks: 570 cycles of pivot arithmetic + two matrix products at "
                            "102 cycles result-to-operand; row solve 2 products per block; back-substitution 312 cycles per block), times the LM trials: the floor of the level-by-level factorisation "
                            "on this chip whatever the launch structure -- the measured stage is `roofline.ms_per_step`"}
    nkp = [pipe.ctx.features_get(f)[0].shape[0] for f in mine[:8]]
    tot_rows, tot_kp7 = pipe.ctx.match_total()
    active_pairs = sum(1 for p in range(len(pipe.src)) if pipe.ctx.pair_is_active(p))
    pairs_per_rank, active_per_rank = [len(pipe.src)], [active_pairs]
    if world > 1:       # pairs go to the owner of their TARGET frame (contiguous blocks): rank r holds ~(2r+1)/world^2 of the dense pair list, but the
        t = torch.zeros((2, world), dtype=torch.int64, device="cuda"); t[0, rank] = len(pipe.src); t[1, rank] = active_pairs   # ACTIVE pairs (the only ones that cost anything) follow the survey's geometry
        dist.all_reduce(t); pairs_per_rank, active_per_rank = t[0].tolist(), t[1].tolist(); active_pairs = int(sum(active_per_rank))

    # ---- the same steps with the raw frames in page-locked HOST memory: the library streams them in under the extraction kernels
    # (SURVEY.md 8d lists the upload inside the metric; `value` stays the HBM-resident figure, this is the PCIe-inclusive one)
    pcie = None
    h_raws = None
    if args.pcie_steps > 0:
        h_raws = [None] * F
        for f in mine:
            h_raws[f] = raws[f].cpu().pin_memory()
        pipe.run(h_raws, poses, alts, grs)
        barrier()
        t1 = time.perf_counter()
        for _ in range(args.pcie_steps):
            pipe.run(h_raws, poses, alts, grs)
        barrier()
        dt1 = time.perf_counter() - t1
        if world > 1:
            t = torch.tensor([dt1], dtype=torch.float64, device="cuda"); dist.all_reduce(t, op=dist.ReduceOp.MAX); dt1 = float(t.item())
        # one more pass by stages (untimed): what the upload achieves and what is left behind the last copy
        tt = [time.perf_counter()]
        pipe.set_frames(h_raws, poses, alts, grs); pipe.extract(); pipe.ctx.sync(); tt.append(time.perf_counter())
        pipe.match(); pipe.optimize(); tt.append(time.perf_counter())
        bytes_step = float(sum(int(h_raws[f].numel()) * 8 for f in mine))
        pcie = {"value": F * args.pcie_steps / dt1, "unit": "frames/s", "ms_per_step": 1e3 * dt1 / args.pcie_steps, "steps": args.pcie_steps,
                "bytes_per_step": bytes_step, "input": "float64 frames in page-locked host memory, uploaded inside every step (double-buffered under the extraction kernels)",
                "extract_with_upload_ms": 1e3 * (tt[1] - tt[0]), "upload_GBs": bytes_step / (tt[1] - tt[0]) / 1e9, "tail_after_last_copy_ms": 1e3 * (tt[2] - tt[1]),
                "note": "the figure SURVEY.md 8(d) defines (upload inside the metric); floor = bytes_per_step / ~55 GB/s of a page-locked upload + the tail that cannot start before the last frame is in "
                        "(matching, mini-LMs, pose graph); upload_GBs = bytes / (frame set-up + extraction with the upload inside), tail = the rest of the step"}

    # ---- throughput with several surveys in flight (an extra, never `value`): the pose-graph solve of one survey is latency-bound
    # and leaves the chip idle, the extraction of the next survey fills it.  Two contexts (own streams), two host threads, whole
    # steps on the same HBM-resident input; single-survey latency is unchanged (it is `ms_per_step`).
    inflight = None
    if args.jobs_in_flight > 1 and world == 1:
        import threading
        pipes = [pipe] + [Pipeline(F, device=local_rank, nfeatures=wl.get("nfeatures")) for _ in range(args.jobs_in_flight - 1)]
        for p in pipes[1:]:
            p.run(raws, poses, alts, grs)
        barrier()

        stagger = dt / args.steps / len(pipes)           # surveys arrive evenly spaced (inside the timed region): solve of one under the extraction of the next

        def run_job(p, k, nsteps, frames):
            time.sleep(k * stagger)
            for _ in range(nsteps):
                p.run(frames, poses, alts, grs)

        def round_of(nsteps, frames):
            th = [threading.Thread(target=run_job, args=(p, k, nsteps, frames)) for k, p in enumerate(pipes)]
            for t in th:
                t.start()
            for t in th:
                t.join()
            for p in pipes:
                p.ctx.sync()
        round_of(1, raws)                               # untimed: the first concurrent steps pay one-off costs (arena growth, worker threads)
        t2 = time.perf_counter()
        round_of(args.steps, raws)
        dt2 = time.perf_counter() - t2
        nst = args.steps * len(pipes)
        inflight = {"surveys_in_flight": len(pipes), "value": F * nst / dt2, "unit": "frames/s", "ms_per_step": 1e3 * dt2 / nst, "steps": nst,
                    "note": "independent surveys overlapped on one GPU (one context and host thread each); a throughput figure for batch processing, the latency of one survey is ms_per_step"}
        if h_raws is not None:                          # the same with host-resident frames: the upload of one survey runs under the solve of the other
            stagger = 1e-3 * pcie["ms_per_step"] / len(pipes)
            round_of(1, h_raws)
            t3 = time.perf_counter()
            round_of(args.pcie_steps, h_raws)
            dt3 = time.perf_counter() - t3
            nst3 = args.pcie_steps * len(pipes)
            inflight["pcie_inclusive"] = {"value": F * nst3 / dt3, "unit": "frames/s", "ms_per_step": 1e3 * dt3 / nst3, "steps": nst3,
                                          "note": "frames in page-locked host memory, uploaded inside every step; PCIe floor = bytes_per_step / ~55 GB/s"}
        for p in pipes[1:]:
            p.close()
    h_raws = None

    # ---- side leg on several GPUs: ONE WHOLE SURVEY PER GPU, no communication -- what DESIGN.md section 4 "Ranks" recommends for throughput (a
    # fleet's surveys are independent; the sharded step above is the latency of ONE survey and is what `value` reports).  Every rank makes the
    # whole survey (3.3 GB at C3), runs it through a context of its own outside the communicator, and the aggregate is all ranks' frames over
    # the slowest rank's time.  Never fatal: a failure here leaves the line without the leg.
    replicas = None
    if (world > 1 or force_comm) and not args.no_side_legs and not big:
        try:
            all_raws = [raws[f] if raws[f] is not None else sv.frame(f) for f in range(F)]
            solo = Pipeline(F, device=local_rank, nfeatures=wl.get("nfeatures"))
            solo_survey = solo.prepare(all_raws, poses, alts, grs)
            for _ in range(max(1, args.warmup)):
                solo.run(solo_survey)
            solo.ctx.sync(); barrier()
            t4 = time.perf_counter()
            for _ in range(args.steps):
                solo.run(solo_survey)
            solo.ctx.sync(); barrier()
            dt4 = time.perf_counter() - t4
            if world > 1:
                t = torch.tensor([dt4], dtype=torch.float64, device="cuda"); dist.all_reduce(t, op=dist.ReduceOp.MAX); dt4 = float(t.item())
            solo.close(); del all_raws
            replicas = {"value": world * F * args.steps / dt4, "unit": "frames/s", "surveys": world, "ms_per_step": 1e3 * dt4 / args.steps, "scaling": "weak",
                        "note": "one whole survey per GPU, no collective in the data path: the throughput arrangement; `value` above is ONE survey sharded over the GPUs (strong scaling, its latency)"}
        except Exception as e:                          # (the main line must not depend on this leg)
            replicas = {"error": "%s: %s" % (type(e).__name__, e)}

    if rank == 0:
        out = {
            "metric": "sonar frames/sec end-to-end (extract+match+LM solve)",
            "value": F * args.steps / dt, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "u8/int32 (extract, match) + f64 (geo gate, LM)", "data": "synthetic",
            "config": {"workload": wl["name"], "frames": F, "pings": N, "bins": M, "pairs": F * (F - 1) // 2, "active_pairs": active_pairs,
                       "active_pairs_note": "pairs whose geo bounding boxes intersect; the others are provably empty (FEAmatcher.cpp:84) and skipped",
                       "input": "raw float64 frames resident in HBM before the timed region (PCIe-inclusive figure: pcie_inclusive)",
                       "harness": "the argument arrays of dsss_frames_set (ids, sizes, per-frame pointers) are built once before the clock, as a C++ caller holds them; every C-ABI call of a step is timed, on every step; no result or analysis is carried between steps",
                       "pairs_per_rank": pairs_per_rank, "active_pairs_per_rank": active_per_rank, "kp_per_frame": int(np.mean(nkp)) if nkp else 0, "matches_rank0": tot_rows, "lc_problems_rank0": tot_kp7,
                       "pg_stats": [float(s) for s in stats] if stats is not None else None,
                       "parallelism": "contiguous frame blocks over %d rank(s): RCCL all-gather of features, pairs to the owner of the target frame, pose graph sharded with one RCCL all-reduce of the reduced Hessian per LM trial" % world},
            "roofline": roof, "roofline_stages": roof_groups, "roofline_all_kernels": roof_all, "breakdown_ms": breakdown, "work_per_step": work, "pcie_inclusive": pcie, "throughput_surveys_in_flight": inflight,
            "match_allpairs": match_allpairs, "sift128": sift_leg,
        }
        if replicas is not None:
            out["throughput_one_survey_per_gpu"] = replicas
        if crit is not None:      # the dependent-chain floor belongs to the factorisation stage; the headline carries it when that stage IS the headline
            if roof_groups and roof_groups.get("pg_factor"):
                out["roofline_stages"] = dict(roof_groups, pg_factor=dict(roof_groups["pg_factor"], critical_path_floor_ms=crit["ms_per_step"], critical_path=crit))
            if roof is not None and str(roof.get("kernel", "")).startswith("pg_factor"):
                out["roofline"] = dict(roof, critical_path_floor_ms=crit["ms_per_step"], critical_path=crit)
            out["pg_critical_path"] = crit
        if pcie is not None:       # SURVEY.md 8(d) lists the upload INSIDE the metric: this is that figure; `value` is the HBM-resident one the bench contract asks for
            out["survey_8d_figure"] = {"value": pcie["value"], "unit": "frames/s", "ms_per_step": pcie["ms_per_step"], "what": "upload of the raw frames inside every step (pcie_inclusive)"}
        if world > 1 or force_comm:
            cs = pipe.ctx.comm_stats()
            out["comm"] = {"allreduce_bytes_total": cs[2], "calls": cs[3]}
        if args.cpu_frames > 0 and world == 1:
            nf = min(args.cpu_frames, F)
            out["cpu_baseline"], o_poses, o_edges = cpu_baseline(sv, wl, nf)
            out["parity_vs_oracle_on_cpu_sample"] = sample_parity(sv, nf, o_poses, o_edges, local_rank, wl.get("nfeatures"))
            nthr = min(os.cpu_count() or 1, 32)
            if nthr > 1:                                        # the same sample on the host's cores (a reported extra, not the baseline)
                out["cpu_baseline_allcores"] = cpu_baseline(sv, wl, nf, threads=nthr)[0]
        print(json.dumps(out))
    pipe.close()
    if world > 1 or force_comm:
        dist.destroy_process_group()


pmc_mfma_table = ({}, None)
F64_PEAK_TFLOPS = 78.6   # MI355X FP64 vector/matrix peak (AMD spec sheet; v_mfma_f64_16x16x4_f64 measured at 64 cycles: the two peaks coincide)
# ONE measured constant for everything that is priced in vector instructions: the chip-wide wave-instruction issue rate of the
# full-rate class at 8 wavefronts per SIMD, tools/ubench/valu_issue.hip on MI355X (profiles/r03_ubench_valu_issue.txt: v_fma_f32
# 1015 G/s = one per 2.4 nominal cycles per SIMD; v_add_u32 / v_xor_b32 786-844).  The half-rate classes (packed 16-bit min/max,
# v_bcnt, v_dot4, v_perm, three-operand integer ops, every f64 op) issue at 490-570 G/s: a kernel made of them fills at most ~0.5.
VALU_ISSUE_PER_S = 1015e9
# comparisons/s against `lanes x clock / ~30 operations per comparison` -- with the measured issue rate for
# lanes x clock.  (The kernel itself spends 13.4 lane-instructions per gate + Hamming evaluation, PMC; it is NOT priced against that.)
MATCH_PEAK_GEVALS = VALU_ISSUE_PER_S * 64 / 30.0 / 1e9

FLOP_SLOTS = {"pg_acc", "pg_rsu", "pg_diag", "pg_trsm", "pg_bwd", "pg_subtree", "lc"}
MFMA_SLOTS = {"pg_acc", "pg_rsu", "pg_diag", "pg_trsm"}             # kernels whose flops run on v_mfma_f64_16x16x4_f64
# stages = kernel groups of SURVEY.md 8(d); the headline `roofline` is the group with the most GPU time
GROUPS = {
    "extract": ("hbm", ["row_reduce", "pre_misc", "normalize", "pyramid", "fast", "fast_compact", "quadtree", "desc", "filter"]),
    "match": ("valu_int", ["match", "scc", "rows"]),
    "lc": ("valu_f64", ["lc"]),
    "pg_factor": ("mfma", ["pg_subtree", "pg_asm", "pg_diag", "pg_trsm", "pg_acc", "pg_rsu", "pg_bwd"]),
}
GROUP_NOTE = {
    "extract": "SURVEY 8(d): 29.4 N M algorithmic bytes per frame (raw f64 twice, L0, pyramid r/w, FAST reads, blur r/w) over the summed time of the extraction kernels, against HBM",
    "pg_factor": "multifrontal factorisation + solves of the reduced pose-graph system, all kernels of one LM trial (bins, extend-add, panel Cholesky, row solve + trailing update -- fused per tile on most levels --, "
                 "back-substitution): algorithmic f64 flops of one factorisation over their summed time, against the f64 matrix peak.  Latency-bound: a dozen panel levels of four dependent launches per trial at C3 since the chain-order dissection of round 5 (29 levels before), see critical_path_floor_ms",
    "match": "SURVEY 8(d) K9: gate + Hamming evaluations against lanes x clock / 30 operations (lanes x clock = the measured issue rate).  `achieved` counts the evaluations the kernel PERFORMS "
             "(counted on the device in the profiled pass): a geo grid of radius / 2 cells hands a query only the keypoints of the 5 x 5 cells around it -- the same set passes the same f64 gate -- which is "
             "`evaluations_performed_of_reference` of the reference's Na x Nb per directed pair; `reference_evaluations_per_s_G` prices the stage in the reference's own count",
    "lc": "f64 VALU, 3e4 flop per LM iteration and match (SURVEY 8(d) K11)",
}

# profile slot -> kernel name in the rocprofv3 tables under profiles/
SLOT_KERNEL = {"row_reduce": "row_reduce_kernel", "normalize": "normalize_kernel", "pyramid": "resize_kernel", "fast": "fast_cells_kernel",
               "desc": "orient_desc_kernel", "quadtree": "quadtree_kernel", "lc": "lc_kernel", "match": "match_grid_kernel<false>",
               "pg_acc": "pg_front_syrk_kernel", "pg_diag": "pg_front_diag4_kernel", "pg_trsm": "pg_front_trsm2_kernel", "pg_bwd": "pg_front_bwd2_kernel",
               "pg_subtree": "pg_factor_subtree_kernel", "pg_asm": "pg_front_asm_kernel", "pg_rsu": "pg_front_rsu_kernel"}


def _latest(pattern):
    import glob
    hits = sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)))
    return hits[-1] if hits else None


def pmc_traffic(workload):
    """HBM bytes per launch from the committed rocprofv3 PMC passes of the latest round: profiles/rNN_pmc_hbm_traffic_<workload>.csv,
    written by tools/pmc_summary.py (separate FETCH_SIZE and WRITE_SIZE passes; FETCH_SIZE x2 only for the kernels whose reads are
    16-byte-per-lane streams, as MI355X_MICROARCH.md prescribes).  Returns kernel -> (bytes, calibrated)"""
    path = _latest("r*_pmc_hbm_traffic_%s.csv" % workload)
    out = {}
    if path:
        import csv
        with open(path) as fh:
            agg = {}
            for r in csv.DictReader(fh):
                k = r["kernel"][5:] if r["kernel"].startswith("void ") else r["kernel"]
                out[k] = (float(r["fetch_bytes_per_launch_used"]) + float(r["write_bytes_per_launch"]), bool(int(r["fetch_calibrated_x2"])))
                a = agg.setdefault(k.split("<")[0], [0.0, 0.0, True])      # template instances of one kernel: bytes per launch averaged over all their launches
                a[0] += out[k][0] * float(r["dispatches"]); a[1] += float(r["dispatches"]); a[2] = a[2] and out[k][1]
            for k, (sb, n, cal) in agg.items():
                if k not in out and n > 0:
                    out[k] = (sb / n, cal)
    return out


def pmc_valu(workload):
    """vector instructions per launch from the committed counter pass (profiles/rNN_pmc_issue_<workload>.csv, tools/pmc_issue_summary.py)"""
    path = _latest("r*_pmc_issue_%s.csv" % workload)
    out = {}
    if path:
        import csv
        with open(path) as fh:
            agg = {}
            for r in csv.DictReader(fh):
                k = r["kernel"][5:] if r["kernel"].startswith("void ") else r["kernel"]
                out[k] = float(r["SQ_INSTS_VALU_per_launch"])
                a = agg.setdefault(k.split("<")[0], [0.0, 0.0])
                n = float(r.get("dispatches", 1) or 1)
                a[0] += out[k] * n; a[1] += n
            for k, (sv, n) in agg.items():
                if k and k not in out and n > 0:
                    out[k] = sv / n
    return out


def pmc_mfma(workload):
    """MfmaUtil (SQ_VALU_MFMA_BUSY_CYCLES over GRBM_GUI_ACTIVE x all SIMDs, percent) per kernel from the committed counter pass of the latest
    round: profiles/rNN_pmc_mfma_<workload>.csv (tools/pmc_mfma_summary.py) -- north_star asks for "MFMA utilisation on the solve" as evidence"""
    path = _latest("r*_pmc_mfma_%s.csv" % workload)
    out = {}
    if path:
        import csv
        with open(path) as fh:
            agg = {}
            for r in csv.DictReader(fh):
                k = r["kernel"][5:] if r["kernel"].startswith("void ") else r["kernel"]
                out[k] = float(r["MfmaUtil_percent_of_all_1024_SIMDs"])
                a = agg.setdefault(k.split("<")[0], [0.0, 0.0])       # template instances of one kernel (tile sizes): mean weighted by GPU time (launches x active cycles)
                w = float(r["dispatches"]) * float(r["GRBM_GUI_ACTIVE_per_launch"])
                a[0] += out[k] * w; a[1] += w
            for k, (sw, w) in agg.items():
                if k not in out and w > 0:
                    out[k] = sw / w
    return out, (os.path.basename(path) if path else None)


def one_roofline(slot, ms, n, work, traffic, valu=None):
    """one kernel: `frac` = algorithmic bytes (or flops, or SURVEY 8(d)'s comparisons) per launch over the live average launch duration,
    against the hardware peak -- never against the kernel's own instruction count; `valu_issue_util` is reported beside it"""
    per_launch_s = ms * 1e-3 / n
    kern = SLOT_KERNEL.get(slot, "")
    tr = traffic.get(kern) or traffic.get(kern.split("<")[0])
    if slot in FLOP_SLOTS:
        ach = work / n / per_launch_s / 1e12
        r = {"kernel": slot, "bound": "mfma" if slot in MFMA_SLOTS else "valu_f64", "achieved": ach, "peak": F64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach / F64_PEAK_TFLOPS}
    elif slot == "match":
        ach = work / n / per_launch_s / 1e9
        r = {"kernel": slot, "bound": "valu_int", "achieved": ach, "peak": MATCH_PEAK_GEVALS, "unit": "G gate+Hamming evaluations/s", "frac": ach / MATCH_PEAK_GEVALS}
    else:
        ach = work / n / per_launch_s / 1e9
        r = {"kernel": slot, "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS}
    if tr:
        r["hbm_GBs"] = tr[0] / per_launch_s / 1e9
    v = valu.get(kern) or valu.get(kern.split("<")[0]) if valu else None
    if v:
        r["valu_issue_util"] = v / per_launch_s / VALU_ISSUE_PER_S      # measured instruction count (PMC) over the live duration
    r.update({"traffic": tr[0] if tr else None, "traffic_calibrated": tr[1] if tr else None, "launches": n, "avg_launch_us": per_launch_s * 1e6})
    return r


def group_roofline(name, prof, traffic, units, unit_work):
    """a stage of SURVEY.md 8(d) = a group of kernels.  One "launch" of the group = one pass over its unit (`units` per profiled step:
    1 for the per-survey stages, the number of LM trials for the factorisation); duration = the summed event time of its kernels."""
    bound, slots = GROUPS[name]
    live = [k for k in slots if k in prof and prof[k][1] > 0]
    if not live:
        return None
    ms = sum(prof[k][0] for k in live)
    per_unit_s = ms * 1e-3 / units
    work = unit_work if unit_work is not None else sum(prof[k][2] for k in live) / units
    tr, cal = 0.0, True
    for k in live:
        kern = SLOT_KERNEL.get(k, "")
        t = traffic.get(kern) or traffic.get(kern.split("<")[0])
        if t is None:
            cal = False
            continue
        tr += t[0] * prof[k][1] / units
        cal = cal and t[1]
    if bound == "hbm":
        ach, peak, unit = work / per_unit_s / 1e9, HBM_PEAK_GBS, "GB/s"
    elif bound == "valu_int":
        ach, peak, unit = work / per_unit_s / 1e9, MATCH_PEAK_GEVALS, "G gate+Hamming evaluations/s"
    else:
        ach, peak, unit = work / per_unit_s / 1e12, F64_PEAK_TFLOPS, "TFLOP/s"
    out = {"kernel": name + " (" + " + ".join(SLOT_KERNEL.get(k, k) for k in live) + ")", "bound": "hbm" if bound == "hbm" else ("mfma" if bound == "mfma" else bound),
           "achieved": ach, "peak": peak, "unit": unit, "frac": ach / peak, "traffic": tr if tr > 0 else None, "traffic_calibrated": cal if tr > 0 else None,
           "launches": units, "avg_launch_us": per_unit_s * 1e6, "ms_per_step": ms, "note": GROUP_NOTE.get(name)}
    if tr > 0:
        out["hbm_GBs"] = tr / per_unit_s / 1e9       # counter traffic over the live duration: the "rocprof HBM GB/s" north_star asks for on the matcher
    if bound == "mfma":                              # "MFMA utilisation on the solve": per kernel from the committed PMC pass, and their mean weighted by the live kernel times
        mu, src = pmc_mfma_table
        per = {SLOT_KERNEL[k]: mu[SLOT_KERNEL[k]] for k in live if SLOT_KERNEL.get(k) in mu}
        if per:
            tw = sum(prof[k][0] for k in live if SLOT_KERNEL.get(k) in mu)
            out["mfma_util_pct_by_kernel"] = per
            out["mfma_util_pct"] = sum(mu[SLOT_KERNEL[k]] * prof[k][0] for k in live if SLOT_KERNEL.get(k) in mu) / tw if tw > 0 else None
            out["mfma_util_source"] = "profiles/" + src
    return out


def roofline(prof, workload, wl=None, trials=1):
    """headline = the kernel GROUP with the largest accumulated GPU time (SURVEY 8(d) stages); plus every group and every kernel
    that has an algorithmic work figure, each against the hardware peak of what bounds it"""
    global pmc_mfma_table
    traffic = pmc_traffic(workload)
    valu = pmc_valu(workload)
    pmc_mfma_table = pmc_mfma(workload)
    F, N, M = (wl["F"], wl["N"], wl["M"]) if wl else (0, 0, 0)
    groups = {}
    match_alg = None
    if "match" in prof and prof.get("match_done", (0, 0, 0))[2] > 0:      # the matcher is priced by the gate evaluations it PERFORMS (its geo grid skips the cells
        match_alg = prof["match"][2]                                       # out of a query's reach); the reference's Na x Nb per pair is reported beside it
        prof = dict(prof); prof["match"] = (prof["match"][0], prof["match"][1], prof["match_done"][2])
    for g in GROUPS:
        units = max(trials, 1) if g == "pg_factor" else 1
        unit_work = 29.4 * N * M * F if g == "extract" else (prof["match"][2] if g == "match" and "match" in prof else None)
        r = group_roofline(g, prof, traffic, units, unit_work)
        if r:
            groups[g] = r
    if not groups:
        return None, None, None
    best = max(groups, key=lambda g: groups[g]["ms_per_step"])
    cand = {k: v for k, v in prof.items() if k not in ("pg", "pg_comm") and v[1] > 0 and v[2] > 0}
    allr = {k: one_roofline(k, *cand[k], traffic, valu) for k in cand}
    keep = ("bound", "achieved", "peak", "unit", "frac", "avg_launch_us", "traffic", "traffic_calibrated", "hbm_GBs", "launches", "valu_issue_util", "ms_per_step",
            "mfma_util_pct", "mfma_util_pct_by_kernel", "mfma_util_source", "evaluations_performed_of_reference", "reference_evaluations_per_s_G", "note")
    rnd = lambda d: {kk: (round(vv, 5) if isinstance(vv, float) else vv) for kk, vv in d.items() if kk in keep}
    if match_alg:
        for r in (groups.get("match"), allr.get("match")):
            if r:
                r["evaluations_performed_of_reference"] = prof["match"][2] / match_alg
                r["reference_evaluations_per_s_G"] = match_alg / (r["avg_launch_us"] * 1e-6 * r["launches"]) / 1e9 if "ms_per_step" not in r else match_alg / (r["ms_per_step"] * 1e-3) / 1e9
    mu = pmc_mfma_table[0]
    for k, v in allr.items():
        if SLOT_KERNEL.get(k) in mu:
            v["mfma_util_pct"] = mu[SLOT_KERNEL[k]]
    return groups[best], {g: rnd(v) for g, v in groups.items()}, {k: rnd(v) for k, v in allr.items()}


if __name__ == "__main__":
    main()
