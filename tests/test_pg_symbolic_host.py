"""CPU: the pose-graph analysis the device kernels run on (ordering with rank partitions, column structures, bins with
update matrices, multifrontal fronts, extend-add maps, schedule -- diasss_amd/csrc/dsss_pg_sym.cpp) pinned through its host
twin: the multifrontal solve of a random SPD block system must equal numpy's dense solve, for one rank and for a rank
partition (where the interface columns are ordered last), for chain-only graphs, dense-ish graphs and lawn-mower-like ones."""
import ctypes as C
import os
import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _system(ns, chords, seed):
    rng = np.random.default_rng(seed)
    ea = list(range(ns - 1)) + [c[0] for c in chords]
    eb = list(range(1, ns)) + [c[1] for c in chords]
    ne = len(ea)
    aval = np.zeros((ns + ne, 36))
    A = np.zeros((6 * ns, 6 * ns))
    for e in range(ne):
        B = rng.normal(0, 0.3, (6, 6))
        aval[ns + e] = B.ravel()
        a, b = ea[e], eb[e]
        A[6 * a:6 * a + 6, 6 * b:6 * b + 6] += B; A[6 * b:6 * b + 6, 6 * a:6 * a + 6] += B.T
    deg = np.zeros(ns)
    for a, b in zip(ea, eb):
        deg[a] += 1; deg[b] += 1
    for k in range(ns):
        Q = rng.normal(0, 1, (6, 6)); D = Q @ Q.T * 0.1 + np.eye(6) * (2.5 * deg[k] + 1.0)
        aval[k] = D.ravel(); A[6 * k:6 * k + 6, 6 * k:6 * k + 6] += D
    rhs = rng.normal(0, 1, (ns, 6))
    return np.array(ea, np.int32), np.array(eb, np.int32), aval, rhs, A


def _solve(ea, eb, aval, rhs, cx, cy, part=None, nparts=1):
    from diasss_amd import capi
    L = capi.lib()
    ns = len(rhs)
    x = np.zeros((ns, 6)); st = np.zeros(8, np.int64)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    rc = L.dsss_host_pg_solve(ns, p(ea), p(eb), len(ea), p(cx), p(cy), p(part) if part is not None else None, nparts,
                              p(np.ascontiguousarray(aval)), p(np.ascontiguousarray(rhs)), p(x), p(st))
    assert rc == 0, rc
    return x, st


def _lawnmower(legs, per_leg, seed, density=0.5):
    """legs x per_leg nodes on a serpentine chain, chords between neighbouring legs at the same along-track position"""
    rng = np.random.default_rng(seed)
    ns = legs * per_leg
    cx = np.zeros(ns); cy = np.zeros(ns)
    for l in range(legs):
        for i in range(per_leg):
            k = l * per_leg + i
            cx[k] = i if l % 2 == 0 else per_leg - 1 - i
            cy[k] = 3.0 * l + 0.1 * rng.standard_normal()
    chords, used = [], set()
    for l in range(legs):
        for dl in (1, 2):
            if l + dl >= legs:
                continue
            for i in range(per_leg):
                if rng.uniform() > density / dl:
                    continue
                j = (per_leg - 1 - i if dl == 1 else i) + int(rng.integers(-1, 2))
                j = min(max(j, 0), per_leg - 1)
                b = (l + dl) * per_leg + j
                if b in used:
                    continue                                           # at most one loop closure per target node
                used.add(b); chords.append((l * per_leg + i, b))
    return ns, chords, cx, cy


@pytest.mark.parametrize("case", ["chain", "random", "lawnmower", "lawnmower_big"])
def test_host_multifrontal_solve_equals_dense(case):
    if case == "chain":
        ns, chords = 40, []
        cx = np.arange(ns, dtype=float); cy = np.zeros(ns)
    elif case == "random":
        ns = 120
        rng = np.random.default_rng(3)
        chords = [(int(a), int(b)) for a, b in rng.integers(0, ns, (150, 2)) if abs(a - b) > 1]
        chords = list({(min(a, b), max(a, b)) for a, b in chords})
        cx = rng.uniform(0, 10, ns); cy = rng.uniform(0, 10, ns)
    elif case == "lawnmower":
        ns, chords, cx, cy = _lawnmower(6, 40, 5)
    else:
        ns, chords, cx, cy = _lawnmower(12, 90, 6)
    ea, eb, aval, rhs, A = _system(ns, chords, 7)
    ref = np.linalg.solve(A, rhs.ravel()).reshape(ns, 6)
    x, st = _solve(ea, eb, aval, rhs, cx, cy)
    assert np.abs(x - ref).max() < 1e-10 * max(1.0, np.abs(ref).max())
    if case == "lawnmower_big":
        assert st[1] > 3 and st[6] > 0 and st[3] >= 2            # fronts, binned columns and several levels all occur


def test_nested_dissection_picks_the_cheaper_axis_on_a_strip(monkeypatch):
    """A survey is a long strip of legs: cutting across the longer extent (between legs) costs the loop closures of two legs
    at every level, cutting across the legs one node per leg.  With both median cuts tried (pg_sym_opts::nd_both_axes) the
    factor of a 40-leg strip must be markedly sparser than with the longer-extent rule alone, and the solve stays exact."""
    legs, per_leg = 40, 30
    ns, chords, cx, cy = _lawnmower(legs, per_leg, 21, density=0.9)
    cy = cy * 4.0                                                # legs 12 apart, 30 long: the strip is 16 x longer than wide
    ea, eb, aval, rhs, A = _system(ns, chords, 23)
    ref = np.linalg.solve(A, rhs.ravel()).reshape(ns, 6)
    monkeypatch.setenv("DSSS_PG_ND_BOTH", "1000000")
    x0, st0 = _solve(ea, eb, aval, rhs, cx, cy)
    monkeypatch.delenv("DSSS_PG_ND_BOTH")
    x1, st1 = _solve(ea, eb, aval, rhs, cx, cy)
    for x in (x0, x1):
        assert np.abs(x - ref).max() < 1e-10 * max(1.0, np.abs(ref).max())
    assert st1[0] < 0.8 * st0[0], (st0.tolist(), st1.tolist())   # factor blocks
    x2, st2 = _solve(ea, eb, aval, rhs, cx, cy)
    assert np.array_equal(x1, x2) and np.array_equal(st1, st2)   # deterministic


@pytest.mark.parametrize("nparts", [2, 3, 4, 8])
def test_host_multifrontal_solve_with_rank_partition(nparts, monkeypatch):
    """contiguous leg blocks per rank: interface columns last, comm children exist, same solution"""
    monkeypatch.setenv("DSSS_PG_BIN_COST", "200")
    legs, per_leg = 16, 50
    ns, chords, cx, cy = _lawnmower(legs, per_leg, 9)
    ea, eb, aval, rhs, A = _system(ns, chords, 11)
    ref = np.linalg.solve(A, rhs.ravel()).reshape(ns, 6)
    part = (np.arange(ns) // per_leg * nparts // legs).astype(np.int32)
    x, st = _solve(ea, eb, aval, rhs, cx, cy, part, nparts)
    assert np.abs(x - ref).max() < 1e-10 * max(1.0, np.abs(ref).max())
    assert st[5] > 0                                             # update matrices cross from interiors into the interface
    x1, st1 = _solve(ea, eb, aval, rhs, cx, cy)
    assert np.abs(x - x1).max() < 1e-9


def test_partitioned_analysis_keeps_the_ownership_invariant(monkeypatch):
    """One leg per rank and chords over one and two legs: a pose is the target of a loop closure from a lower rank AND the source
    of one into a higher rank, so its lower-rank partner can lose sight of it (it leaves into a separator higher up) before a
    cut runs between the two.  A factor belongs to the rank of its higher pose and adds to the diagonal block of the lower one:
    that lower node must be interface whatever the cuts were (dsss_host_pg_solve fails with DSSS_E_STATE otherwise); the solve stays exact."""
    monkeypatch.setenv("DSSS_PG_BIN_COST", "200")
    legs, per_leg = 6, 60
    ns, chords, cx, cy = _lawnmower(legs, per_leg, 31, density=0.8)
    ea, eb, aval, rhs, A = _system(ns, chords, 33)
    ref = np.linalg.solve(A, rhs.ravel()).reshape(ns, 6)
    for nparts in (3, 6):
        part = (np.arange(ns) // per_leg * nparts // legs).astype(np.int32)
        x, st = _solve(ea, eb, aval, rhs, cx, cy, part, nparts)           # rc == 0 is asserted inside
        assert np.abs(x - ref).max() < 1e-10 * max(1.0, np.abs(ref).max())
        assert st[5] > 0


def _solve_local(ea, eb, aval, rhs, cx, cy, last):
    from diasss_amd import capi
    L = capi.lib()
    ns = len(rhs)
    x = np.zeros((ns, 6)); st = np.zeros(8, np.int64)
    last = np.ascontiguousarray(last, np.int32)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    rc = L.dsss_host_pg_solve_local(ns, p(ea), p(eb), len(ea), p(cx), p(cy), p(last), len(last),
                                    p(np.ascontiguousarray(aval)), p(np.ascontiguousarray(rhs)), p(x), p(st))
    return rc, x, st


@pytest.mark.parametrize("case", ["none", "few", "scattered", "all_but_one", "rank_view"])
def test_host_solve_with_a_prescribed_interface(case, monkeypatch):
    """The analysis ONE RANK of several runs (dsss_pg.hip, rank-local mode; pg_sym_opts::iface_last): the listed interface nodes are
    eliminated last as one dense front whatever their edges (checked inside dsss_host_pg_solve_local: last front, exactly these
    columns in this order, no packed update matrices), everything else is ordered by the nested dissection; the solve stays exact.
    `rank_view`: the graph rank 1 of 3 sees -- its own legs + the interface nodes of the whole survey, which are isolated from
    each other and from the chain except through its own nodes."""
    monkeypatch.setenv("DSSS_PG_BIN_COST", "200")
    legs, per_leg = 9, 40
    ns, chords, cx, cy = _lawnmower(legs, per_leg, 41, density=0.7)
    ea, eb, aval, rhs, A = _system(ns, chords, 43)
    rng = np.random.default_rng(5)
    if case == "rank_view":
        part = np.arange(ns) // per_leg * 3 // legs
        forced = np.zeros(ns, bool)
        for a, b in zip(ea, eb):
            if part[a] < part[b]: forced[a] = True
            if part[b] < part[a]: forced[b] = True
        keep = forced | (part == 1)
        loc = -np.ones(ns, int); loc[keep] = np.arange(keep.sum())
        sel = [e for e in range(len(ea)) if keep[ea[e]] and keep[eb[e]]]
        nl = int(keep.sum())
        ea2 = np.array([loc[ea[e]] for e in sel], np.int32); eb2 = np.array([loc[eb[e]] for e in sel], np.int32)
        aval2 = np.concatenate([aval[:ns][keep], aval[ns:][sel]])
        A2 = np.zeros((6 * nl, 6 * nl))
        for k in range(nl):
            A2[6 * k:6 * k + 6, 6 * k:6 * k + 6] = aval2[k].reshape(6, 6)
        for e in range(len(sel)):
            B = aval2[nl + e].reshape(6, 6); a, b = ea2[e], eb2[e]
            A2[6 * a:6 * a + 6, 6 * b:6 * b + 6] += B; A2[6 * b:6 * b + 6, 6 * a:6 * a + 6] += B.T
        last = loc[np.nonzero(forced)[0]]
        assert 0 < len(last) < nl - 10
        rhs2 = rhs[keep]
        ref = np.linalg.solve(A2, rhs2.ravel()).reshape(nl, 6)
        rc, x, st = _solve_local(ea2, eb2, aval2, rhs2, cx[keep], cy[keep], last)
        assert rc == 0 and np.abs(x - ref).max() < 1e-10 * max(1.0, np.abs(ref).max())
        assert st[5] >= len(last) and st[6] > 0                  # interface values exist (at least the diagonals); bins too
        return
    last = {"none": [], "few": [17, 18, 200], "scattered": sorted(rng.choice(ns, 40, replace=False).tolist()),
            "all_but_one": [k for k in range(ns) if k != 77]}[case]
    ref = np.linalg.solve(A, rhs.ravel()).reshape(ns, 6)
    rc, x, st = _solve_local(ea, eb, aval, rhs, cx, cy, last)
    assert rc == 0, rc
    assert np.abs(x - ref).max() < 1e-10 * max(1.0, np.abs(ref).max())
    if last:
        assert st[7] >= len(last)
    # not ascending / out of range: refused
    if case == "few":
        assert _solve_local(ea, eb, aval, rhs, cx, cy, [18, 17])[0] != 0
        assert _solve_local(ea, eb, aval, rhs, cx, cy, [ns])[0] != 0


@pytest.mark.parametrize("case,K", [("chain", 2), ("chain", 5), ("random", 3), ("lawnmower", 2), ("lawnmower", 4), ("lawnmower_big", 8), ("lawnmower_big", 16)])
def test_host_solve_analysed_by_parts(case, K, monkeypatch):
    """ONE rank, analysed by parts (pg_symbolic_parts, round 6): K parts of the chain order are ordered and analysed independently -- all
    at the same time on the worker pool -- with the interface between the parts as the last dense front, and the K sets of tables are joined
    into one (columns, factor positions, bins, roots, fronts, children, original entries, destinations: every index space behind the
    previous part's).  The joined tables must solve the system exactly as the dense solver does, and twice the same way."""
    monkeypatch.setenv("DSSS_PG_BIN_COST", "200")
    if case == "chain":
        ns, chords = 60, []
        cx = np.arange(ns, dtype=float); cy = np.zeros(ns)
    elif case == "random":
        ns = 120
        rng = np.random.default_rng(3)
        chords = [(int(a), int(b)) for a, b in rng.integers(0, ns, (150, 2)) if abs(a - b) > 1]
        chords = list({(min(a, b), max(a, b)) for a, b in chords})
        cx = rng.uniform(0, 10, ns); cy = rng.uniform(0, 10, ns)
    elif case == "lawnmower":
        ns, chords, cx, cy = _lawnmower(6, 40, 5)
    else:
        ns, chords, cx, cy = _lawnmower(16, 90, 6)
    ea, eb, aval, rhs, A = _system(ns, chords, 7)
    ref = np.linalg.solve(A, rhs.ravel()).reshape(ns, 6)
    from diasss_amd import capi
    L = capi.lib()
    p = lambda a: a.ctypes.data_as(C.c_void_p)

    def run():
        x = np.zeros((ns, 6)); st = np.zeros(8, np.int64)
        rc = L.dsss_host_pg_solve_parts(ns, p(ea), p(eb), len(ea), p(cx), p(cy), K, p(np.ascontiguousarray(aval)), p(np.ascontiguousarray(rhs)), p(x), p(st))
        assert rc == 0, rc
        return x, st
    x, st = run()
    assert np.abs(x - ref).max() < 1e-10 * max(1.0, np.abs(ref).max())
    x2, st2 = run()
    assert np.array_equal(x, x2) and np.array_equal(st, st2)     # deterministic whatever the threads do
    if case == "lawnmower_big":
        assert st[1] > 3 and st[6] > 0 and st[3] >= 2            # fronts, binned columns and several levels all occur


def test_host_solver_under_thread_sanitizer():
    """the analysis runs its parallel phases on a process-wide worker pool (spin-then-sleep workers, stolen-back tasks):
    three threads solve different graphs at once under -fsanitize=thread (tools/sanitize/run.sh; ASan/UBSan: `run.sh asan`)"""
    import shutil, subprocess
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    probe = subprocess.run("echo 'int main(){}' | g++ -x c++ -fsanitize=thread - -o /dev/null", shell=True, capture_output=True)
    if probe.returncode != 0:
        pytest.skip("g++ without libtsan")
    out = subprocess.run(["bash", os.path.join(ROOT, "tools", "sanitize", "run.sh"), "tsan"], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "sanitizers: clean" in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]
