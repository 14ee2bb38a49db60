#!/usr/bin/env python3
"""Write a synthetic survey for diasss_amd/host/test_demo, either as flat binary frame dumps

    frame_%03d.bin = int32 N, int32 M, f64 raw[N*M], f64 pose[N*6], f64 alt[N], f64 gr[M/2]

or (--reference-layout) in the reference's own input layout (src/util/util.cpp:45-213): five folders, OpenCV
FileStorage matrices "ct_img" / "auv_pose" / "anno_kps" (XML, or YAML with --yaml) and one-number-per-line text files
for altitude and ground range.

    python tools/export_survey.py OUT F N M SEED [--reference-layout] [--yaml]
"""
import os
import sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def export(out_dir, F=3, N=700, M=480, seed=77):
    from diasss_amd.synth import Survey
    os.makedirs(out_dir, exist_ok=True)
    sv = Survey(F, N, M, seed=seed)
    for f in range(F):
        raw = sv.frame(f).cpu().numpy()
        pose, alt, gr = sv.inputs(f)
        with open(os.path.join(out_dir, "frame_%03d.bin" % f), "wb") as fh:
            fh.write(np.array([N, M], np.int32).tobytes())
            for a in (raw, pose, alt, gr):
                fh.write(np.ascontiguousarray(a, np.float64).tobytes())
    return sv


_DT = {np.dtype("float64"): "d", np.dtype("int32"): "i", np.dtype("uint8"): "u"}


def _num(v, dt):
    return repr(float(v)) if dt == "d" else str(int(v))          # repr(float) round-trips exactly


def write_storage(path, node, mat, yaml=False):
    """one matrix in cv::FileStorage form (the subset filestorage.cpp reads and OpenCV writes)"""
    mat = np.ascontiguousarray(mat)
    dt = _DT[mat.dtype]
    rows, cols = (mat.shape + (1,))[:2] if mat.ndim else (1, 1)
    vals = [_num(v, dt) for v in mat.reshape(-1)]
    with open(path, "w") as fh:
        if yaml:
            fh.write("%YAML:1.0\n---\n" + node + ": !!opencv-matrix\n   rows: %d\n   cols: %d\n   dt: %s\n   data: [ " % (rows, cols, dt))
            for i in range(0, len(vals), 6):
                fh.write(", ".join(vals[i:i + 6]) + (",\n       " if i + 6 < len(vals) else " ]\n"))
            if not vals:
                fh.write("]\n")
        else:
            fh.write('<?xml version="1.0"?>\n<opencv_storage>\n<%s type_id="opencv-matrix">\n  <rows>%d</rows>\n  <cols>%d</cols>\n  <dt>%s</dt>\n  <data>\n' % (node, rows, cols, dt))
            for i in range(0, len(vals), 6):
                fh.write("    " + " ".join(vals[i:i + 6]) + "\n")
            fh.write("  </data></%s>\n</opencv_storage>\n" % node)


def export_reference_layout(out_dir, frames, yaml=False):
    """frames: list of (raw N x M f64, pose N x 6 f64, alt N, gr M/2, anno K x 7 int32 or None)"""
    ext = ".yml" if yaml else ".xml"
    sub = {k: os.path.join(out_dir, k) for k in ("image", "pose", "altitude", "groundrange", "annotation")}
    for d in sub.values():
        os.makedirs(d, exist_ok=True)
    for f, (raw, pose, alt, gr, anno) in enumerate(frames):
        write_storage(os.path.join(sub["image"], "img_%03d%s" % (f, ext)), "ct_img", np.asarray(raw, np.float64), yaml)
        write_storage(os.path.join(sub["pose"], "pose_%03d%s" % (f, ext)), "auv_pose", np.asarray(pose, np.float64), yaml)
        np.savetxt(os.path.join(sub["altitude"], "alt_%03d.txt" % f), np.asarray(alt, np.float64), fmt="%.17g")
        np.savetxt(os.path.join(sub["groundrange"], "gr_%03d.txt" % f), np.asarray(gr, np.float64), fmt="%.17g")
        if anno is not None:
            write_storage(os.path.join(sub["annotation"], "anno_%03d%s" % (f, ext)), "anno_kps", np.asarray(anno, np.int32), yaml)
    return sub


if __name__ == "__main__":
    a = [v for v in sys.argv if not v.startswith("--")]
    if "--reference-layout" in sys.argv:
        from diasss_amd.synth import Survey
        F, N, M, seed = (int(v) for v in a[2:6])
        sv = Survey(F, N, M, seed=seed)
        export_reference_layout(a[1], [(sv.frame(f).cpu().numpy(),) + tuple(sv.inputs(f)) + (None,) for f in range(F)], yaml="--yaml" in sys.argv)
    else:
        export(a[1], *(int(v) for v in a[2:6]))
