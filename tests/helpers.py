"""shared builders for the parity tests (seeded, small)"""
import numpy as np
from oracle import binding as O


def track(N, M, leg, res=0.05, spacing=None, seed=0):
    """DR pose / altitude / ground range of one straight leg (even legs head +x, odd legs -x)"""
    rng = np.random.default_rng(1000 + seed + leg)
    half = M // 2
    gr = res * np.arange(half, dtype=np.float64)
    spacing = spacing if spacing is not None else 0.39 * 2 * half * res
    s = (np.arange(N) + 0.5) * res
    fwd = leg % 2 == 0
    pose = np.zeros((N, 6))
    pose[:, 2] = (0.0 if fwd else 3.14159265359) + 0.002 * rng.standard_normal()
    pose[:, 3] = (s if fwd else N * res - s) + 0.01 * rng.standard_normal(N).cumsum() * 0.1
    pose[:, 4] = leg * spacing + 0.02 * np.sin(s)
    alt = 9.0 + np.sin(s / 7.0)
    return pose, alt, gr


def random_features(N, M, n, seed, margin=100):
    """n keypoints with integer level-0-like coordinates and random descriptors"""
    rng = np.random.default_rng(seed)
    kps = np.zeros(n, O.KP_DTYPE)
    kps["y"] = rng.integers(margin, N - margin, n).astype(np.float32) + rng.choice([0.0, 0.25, 0.5], n).astype(np.float32)
    xs = rng.integers(margin // 2, M - margin // 2, n)
    kps["x"] = xs.astype(np.float32)
    kps["size"] = 31; kps["angle"] = rng.uniform(0, 360, n).astype(np.float32); kps["response"] = rng.integers(7, 200, n)
    kps["octave"] = rng.integers(0, 6, n)
    desc = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    return kps, desc


def paired_features(N, M, n, seed, flip_bits=12, reverse=False, margin=100):
    """frame A random; frame B sees the same 'landmarks' (descriptor + a few flipped bits) at the image position
    that maps to (almost) the same geo point when B runs the neighbouring leg"""
    rng = np.random.default_rng(seed + 7)
    ka, da = random_features(N, M, n, seed, margin)
    kb = ka.copy(); db = da.copy()
    for i in range(n):
        bits = rng.choice(256, rng.integers(0, flip_bits + 1), replace=False)
        for b in bits:
            db[i, b // 8] ^= np.uint8(1 << (b % 8))
    return ka, da, kb, db
