#!/usr/bin/env python3
"""What ONE rank of a W-rank job does, timed on a single GPU (no W-GPU node is available to a builder).

  1. record   all W ranks of the sharded pipeline run in LOCK STEP inside this process -- one context and one host thread per
              rank, the library's host-callback transport (dsss_comm_init_callback) summing / gathering across the threads --
              and every rank keeps the RESULT of each collective it took part in (the all-gathered features, the exchanged
              loop closures, the all-reduced interface buffer of every LM trial, the trajectory).
  2. replay   rank r runs ALONE as rank r of W through the device-callback transport (dsss_comm_init_device_callback); a
              collective copies the recorded result into the device buffer (device to device, a few microseconds) instead of
              talking to anybody.  The step time is then that rank's critical path minus the wire time of the collectives.

The sequence of collectives of a step is a function of the inputs alone, so the recording of one step replays every step.
Implied W-GPU step = max over ranks of the replayed step + the wire time of its collectives (bytes / per-link xGMI rate).

    python tools/emulate_ranks.py [world=8] [workload=C3]          (also: bench.py --emulate-rank r/W | all/W)
"""
import ctypes
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


class LockStep:
    """in-process transport for W threads: op 0 sums float64 buffers, op 1 gathers byte rows; every rank records what it got"""

    def __init__(self, world):
        self.world = world
        self.bar = threading.Barrier(world)
        self.slot = [None] * world
        self.log = [[] for _ in range(world)]          # per rank: (op, result array)
        self.tmp = None

    def fn(self, rank):
        def cb(op, arr):
            self.slot[rank] = arr
            self.bar.wait()
            if op == 0:
                if rank == 0:                           # one sum, in rank order: every rank gets identical bits
                    acc = self.slot[0].copy()
                    for r in range(1, self.world):
                        acc += self.slot[r]
                    self.tmp = acc
                self.bar.wait()
                arr[:] = self.tmp
            else:
                if rank == 0:
                    self.tmp = np.stack([self.slot[r][r] for r in range(self.world)])
                self.bar.wait()
                arr[:] = self.tmp
            self.log[rank].append((op, self.tmp))       # every rank receives the same result: ONE copy is kept (C5: 3 GB of collectives per step)
            self.bar.wait()
        return cb


class Replay:
    """device-callback transport of ONE rank: the k-th collective of a step receives the recorded result of the k-th collective"""

    def __init__(self, log, device):
        import torch
        from diasss_amd import capi
        capi.lib()
        self.hip = capi._HIP                            # the process-wide HIP runtime the library runs on
        self.hip.hipMemcpyAsync.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]
        self.dev = [(op, torch.from_numpy(a.view(np.uint8).reshape(-1)).to(device)) for op, a in log]
        self.bytes = [int(t.numel()) for _, t in self.dev]
        self.k = 0
        self.wire_bytes = 0.0

    def begin_step(self):
        self.k = 0

    def fn(self, op, ptr, n, stream):
        rop, t = self.dev[self.k]
        assert rop == op and (t.numel() == n * 8 if op == 0 else t.numel() % n == 0), "collective sequence differs from the recording"
        rc = self.hip.hipMemcpyAsync(ptr, t.data_ptr(), t.numel(), 3, stream)        # hipMemcpyDeviceToDevice
        assert rc == 0
        self.k += 1


def is_big(wl):
    return wl["F"] * wl["N"] * wl["M"] > (1 << 32)          # C5: 65 GB of raw frames (bench.py: device-side noise, frames of one rank at a time here)


def inputs(wl, seed, device, frames):
    from diasss_amd.synth import Survey
    F, N, M = wl["F"], wl["N"], wl["M"]
    sv = Survey(F, N, M, seed=seed, device=device, noise_on_device=is_big(wl))
    raws = [sv.frame(f) if f in frames else None for f in range(F)]
    ins = [sv.inputs(f) for f in range(F)]
    return raws, [i[0] for i in ins], [i[1] for i in ins], [i[2] for i in ins]


def record(wl, seed, world, device=0):
    """lock-step run of all ranks; returns (per-rank collective logs, the trajectory of rank 0, LM stats, the inputs).  A big workload
    (C5) runs stage by stage: the raw frames of all ranks are resident only until every rank has extracted (65 GB), the pose-graph
    stage of the W contexts then has the device to itself; the replay regenerates one rank's frames at a time."""
    import torch
    from diasss_amd.pipeline import Pipeline
    F = wl["F"]
    big = is_big(wl)
    if big and world > 4:                                    # the extraction scratch of one context is bounded at 24 GB: eight of them plus 65 GB of frames do not fit one device
        os.environ.setdefault("DSSS_EX_SCRATCH_MB", str(max(2048, 65536 // world)))
    raws, poses, alts, grs = inputs(wl, seed, "cuda:%d" % device, set(range(F)))
    ls = LockStep(world)
    pipes = []
    for r in range(world):
        p = Pipeline(F, device=device, rank=r, world=world, nfeatures=wl.get("nfeatures"))
        p.ctx.comm_init_callback(r, world, ls.fn(r))
        pipes.append(p)
    out = [None] * world
    err = []
    stage_bar = threading.Barrier(world)

    def run(r):
        try:
            mine = set(range(F * r // world, F * (r + 1) // world))
            my = [x if f in mine else None for f, x in enumerate(raws)]
            if not big:
                res = pipes[r].run(my, poses, alts, grs)
            else:
                pipes[r].set_frames(my, poses, alts, grs); pipes[r].extract(); pipes[r].ctx.sync()
                del my
                keep = getattr(pipes[r].ctx, "_keep", {})   # (the binding keeps the borrowed raw tensors alive; nothing reads them after the extraction)
                for k in list(keep):
                    keep[k] = (None,) + tuple(keep[k][1:])
                if stage_bar.wait() == 0:                # one thread lets the raw frames go once every rank has extracted
                    for f in range(F):
                        raws[f] = None
                    torch.cuda.empty_cache()
                stage_bar.wait()
                pipes[r].match(); res = pipes[r].optimize()
            out[r] = (res[0].copy(), np.array(res[1]))
        except Exception as e:                      # a failed rank must not leave the others in the barrier
            err.append(e); ls.bar.abort(); stage_bar.abort()
    th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in th: t.start()
    for t in th: t.join()
    if err:
        raise err[0]
    torch.cuda.synchronize()
    free_b, total_b = torch.cuda.mem_get_info()
    print("[emulate] lock-step run of %d ranks done: %.0f of %.0f GB of device memory in use" % (world, (total_b - free_b) / 2**30, total_b / 2**30), file=sys.stderr, flush=True)
    for p in pipes:
        p.close()
    torch.cuda.empty_cache()
    for r in range(1, world):
        assert (out[r][0] == out[0][0]).all(), "ranks disagree"
    return ls.log, out[0][0], out[0][1], (None if big else raws, poses, alts, grs)


def replay_rank(wl, world, r, log, data, steps=5, warmup=2, device=0, profile=False):
    """rank r alone; returns ms per step, the collective bytes of a step, and (optionally) the kernel-family breakdown"""
    import torch
    from diasss_amd.pipeline import Pipeline
    F = wl["F"]
    raws, poses, alts, grs = data
    mine = set(range(F * r // world, F * (r + 1) // world))
    if raws is None:                                    # big workload: this rank's frames only, made again (same seed, same generator: the same frames)
        my_raws = inputs(wl, data_seed(wl), "cuda:%d" % device, mine)[0]
    else:
        my_raws = [x if f in mine else None for f, x in enumerate(raws)]
    rp = Replay(log, "cuda:%d" % device)
    pipe = Pipeline(F, device=device, rank=r, world=world, nfeatures=wl.get("nfeatures"))
    pipe.ctx.comm_init_device_callback(r, world, rp.fn)
    res = None
    # the argument arrays of dsss_frames_set, built once before the clock exactly as bench.py's single-GPU line does (Pipeline.prepare:
    # what a C++ caller of the C ABI holds anyway); rounds 3 - 5 rebuilt them in Python inside every timed step of the replay: 1.7 of a
    # rank's 2.2 ms "frames" stage
    survey = pipe.prepare(my_raws, poses, alts, grs)
    for _ in range(warmup):
        rp.begin_step(); res = pipe.run(survey)
    torch.cuda.synchronize(); pipe.ctx.sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        rp.begin_step(); res = pipe.run(survey)
    torch.cuda.synchronize(); pipe.ctx.sync()
    ms = 1e3 * (time.perf_counter() - t0) / steps
    brk = None
    if profile:
        pipe.ctx.profile(True); pipe.ctx.profile_reset()
        rp.begin_step(); pipe.run(survey)
        pipe.ctx.sync()
        brk = {k: round(v[0], 3) for k, v in pipe.ctx.profile_get().items() if v[1] > 0}
        pipe.ctx.profile(False)
        # wall time of the four host-visible stages (each closed by a stream synchronisation): frames, extraction + feature
        # all-gather, matching + mini-LMs, pose graph (selection, edge exchange, solve, trajectory all-reduce)
        rp.begin_step()
        st = {}
        for name, fn in (("frames", lambda: pipe.set_frames(survey)), ("extract+allgather", pipe.extract), ("match+lc", pipe.match), ("posegraph", pipe.optimize)):
            t1 = time.perf_counter(); fn(); pipe.ctx.sync(); st[name] = round(1e3 * (time.perf_counter() - t1), 3)
        brk["stage_wall_ms"] = st
    out = res[0].copy()
    pipe.close()
    # wire model: ring all-reduce moves 2 (W-1)/W of the buffer over every link, all-gather (W-1)/W of the total
    ar = sum(b for (op, _), b in zip(log, rp.bytes) if op == 0); ag = sum(b for (op, _), b in zip(log, rp.bytes) if op == 1)
    return ms, {"allreduce_bytes": ar, "allgather_bytes": ag, "collectives": len(log)}, brk, out


XGMI_LINK_GBS = 153.0     # per link and direction (task statement): a ring step is bound by ONE link


def wire_ms(comm, world):
    return 1e3 * (2.0 * (world - 1) / world * comm["allreduce_bytes"] + (world - 1) / world * comm["allgather_bytes"]) / (XGMI_LINK_GBS * 1e9)


_SEED = {}


def data_seed(wl):
    return _SEED[wl["name"]]


def emulate(wl, seed, world, ranks=None, steps=5, warmup=2, device=0, profile=False):
    _SEED[wl["name"]] = seed
    log, traj, stats, data = record(wl, seed, world, device)
    ranks = list(range(world)) if ranks is None else ranks
    rows = []
    for r in ranks:
        ms, comm, brk, out = replay_rank(wl, world, r, log[r], data, steps, warmup, device, profile)
        assert (out == traj).all(), "replayed rank %d does not reproduce the lock-step trajectory" % r
        rows.append({"rank": r, "ms_per_step": ms, "wire_ms_model": wire_ms(comm, world), **comm, **({"breakdown_ms": brk} if brk else {})})
    return rows, stats


if __name__ == "__main__":
    import json
    sys.path.insert(0, ROOT)
    from bench import WORKLOADS
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    name = sys.argv[2] if len(sys.argv) > 2 else "C3"
    wl = WORKLOADS[name]
    rows, stats = emulate(wl, 20240601 + ["C2", "C3", "smoke", "C5"].index(name), world, profile=True)
    for r in rows:
        print(json.dumps(r))
    worst = max(r["ms_per_step"] + r["wire_ms_model"] for r in rows)
    print("implied %d-GPU step: %.2f ms (slowest rank + modelled wire time); LM iterations %d" % (world, worst, stats[0]))
