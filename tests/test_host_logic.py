"""CPU tests of the product's host-side logic and of the C-ABI surface (no GPU needed):
- libdsss.so loads and exports every symbol include/dsss.h declares;
- the host quadtree (DistributeOctTree) agrees with the oracle on seeded candidate sets;
- without a device the library fails loudly (no CPU fallback)."""
import ctypes as C
import os
import re
import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "dsss.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(dsss_[a-z0-9_]+)\s*\(", hdr)))


def test_cabi_exports_every_declared_symbol():
    from diasss_amd import capi
    L = capi.lib()
    syms = _declared_symbols()
    assert len(syms) >= 40
    missing = [s for s in syms if not hasattr(L, s)]
    assert not missing, missing


def test_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    from diasss_amd import capi
    with pytest.raises(capi.DsssError):
        capi.Context(max_frames=2)
    assert b"no HIP device" in capi.lib().dsss_strerror(-1)


@pytest.mark.parametrize("seed,n,w,h,quota", [(1, 3000, 603, 903, 300), (2, 20000, 990, 1960, 501), (3, 50, 400, 300, 100),
                                              (4, 5000, 1200, 300, 242), (5, 1, 100, 100, 10), (6, 800, 170, 370, 201)])
def test_host_quadtree_matches_oracle(orc, seed, n, w, h, quota):
    from diasss_amd import capi
    rng = np.random.default_rng(seed)
    pts = set()
    while len(pts) < n:
        pts.add((int(rng.integers(3, w - 3)), int(rng.integers(3, h - 3))))
    pts = sorted(pts, key=lambda p: (p[1] // 30, p[0] // 30, p[1], p[0]))     # cell-major like the FAST output
    xs = np.array([p[0] for p in pts], np.float32); ys = np.array([p[1] for p in pts], np.float32)
    resp = rng.integers(7, 120, n).astype(np.float32)                      # many ties on purpose
    k_o = np.zeros(n, np.int32)
    no = orc.lib().orc_quadtree(orc.fp(xs), orc.fp(ys), orc.fp(resp), n, 16, 16 + w, 16, 16 + h, quota, orc.ip(k_o))
    k_p = np.zeros(n, np.int32); npk = C.c_int(0)
    rc = capi.lib().dsss_host_quadtree(xs.ctypes.data_as(C.c_void_p), ys.ctypes.data_as(C.c_void_p), resp.ctypes.data_as(C.c_void_p), n,
                                       16, 16 + w, 16, 16 + h, quota, k_p.ctypes.data_as(C.c_void_p), C.byref(npk))
    assert rc == 0 and npk.value == no
    assert (k_p[:no] == k_o[:no]).all()
