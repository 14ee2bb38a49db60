"""GPU: the C++ drop-in classes (Frame / FEAmatcher / Optimizer / Util) driven by test_demo, the loop of
src/diasss2.cpp:83-101, must reproduce the oracle's trajectory file."""
import os
import subprocess
import sys
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_test_demo_end_to_end(orc, tmp_path):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import export_survey
    F, N, M = 3, 700, 480
    sv = export_survey.export(str(tmp_path / "frames"), F, N, M, seed=77)
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "diasss_amd", "host")])
    env = dict(os.environ, DSSS_OUT_DIR=str(tmp_path))
    out = subprocess.run([os.path.join(ROOT, "diasss_amd", "host", "test_demo"), str(tmp_path / "frames"), "0.0"],
                         env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    est = np.loadtxt(tmp_path / "est_poses_all.txt")
    drf = np.loadtxt(tmp_path / "dr_poses_all.txt")
    assert est.shape == (F * N, 6) and drf.shape == (F * N, 6)
    # oracle pipeline on the same inputs
    fr = []
    for f in range(F):
        raw = sv.frame(f).numpy(); pose, alt, gr = sv.inputs(f)
        kps, desc, _, _ = orc.detect_feature(raw)
        fr.append(dict(pose=pose, alt=alt, gr=gr, kps=kps, desc=desc, geo=orc.geo_at_kps(pose, gr, M, kps), bb=orc.geo_bbox(pose, gr, M)))
    ps, pt, off, k7, lc = [], [], [0], [], []
    for i in range(F):
        for j in range(i + 1, F):
            a, b = fr[i], fr[j]
            rows = orc.robust_matching(i, j, N, N, a["kps"], a["desc"], a["geo"], a["bb"], b["kps"], b["desc"], b["geo"], b["bb"])
            kp7 = orc.get_kps_pairs(rows, j, a["alt"], a["gr"], b["alt"], b["gr"])
            ps.append(i); pt.append(j); off.append(off[-1] + len(kp7)); k7.append(kp7)
            lc.append(orc.lc_solve(kp7, a["pose"], a["alt"], a["gr"], M, b["pose"], b["alt"], b["gr"], M))
    edges = orc.pg_select_lc([N] * F, ps, pt, off, np.concatenate(k7), np.concatenate(lc))
    o_out, _ = orc.pg_solve(np.concatenate([f["pose"] for f in fr]), edges)
    assert np.abs(est[:, 3:] - o_out[:, 9:]).max() < 2e-6          # 9-decimal text file + 1e-6 solver tolerance
    yaw = np.arctan2(o_out[:, 3], o_out[:, 0])
    dy = np.angle(np.exp(1j * (est[:, 2] - yaw)))
    assert np.abs(dy).max() < 2e-6
    assert np.abs(drf[:, 3:] - np.concatenate([f["pose"] for f in fr])[:, 3:]).max() < 1e-8


def test_test_demo_reference_input_layout(tmp_path):
    """the same survey through the reference's own input layout (five folders, FileStorage XML + txt, Util::LoadInputData)
    must give the same trajectory file as the flat dumps: the loader is exact, so the files agree to the last digit"""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import export_survey
    F, N, M = 3, 700, 480
    sv = export_survey.export(str(tmp_path / "frames"), F, N, M, seed=77)
    d = export_survey.export_reference_layout(str(tmp_path / "ref"), [(sv.frame(f).numpy(),) + tuple(sv.inputs(f)) + (None,) for f in range(F)])
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "diasss_amd", "host")])
    exe = os.path.join(ROOT, "diasss_amd", "host", "test_demo")
    outs = []
    for name, args in (("a", [str(tmp_path / "frames"), "0.0"]),
                       ("b", ["--image", d["image"], "--pose", d["pose"], "--altitude", d["altitude"], "--groundrange", d["groundrange"], "--min-overlap", "0.0"])):
        od = tmp_path / name; od.mkdir()
        out = subprocess.run([exe] + args, env=dict(os.environ, DSSS_OUT_DIR=str(od)), capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stdout + out.stderr
        outs.append(np.loadtxt(od / "est_poses_all.txt"))
    assert outs[0].shape == (F * N, 6) and (outs[0] == outs[1]).all()
