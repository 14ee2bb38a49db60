"""Seeded synthetic side-scan-sonar survey generator.

The reference ships no data (its README points at ../test_data which is not in the tree), so every
benchmark and parity input is synthetic: a lawn-mower AUV track of F parallel legs with alternating
heading 0 / pi (frame-id parity == direction, which is what FEAmatcher.cpp:144,209 assumes), a periodic
procedural seafloor texture, and each leg rendered into an N x M float64 waterfall through the same
ground-range geometry the pipeline assumes (frame.cpp:126-165).  DR poses = truth + drift.

torch is used purely as an array library (CPU here, GPU on the bench box); nothing in here is on the
measured path.
"""
import math
import numpy as np
import torch

PI_REF = 3.14159265359  # the reference's PI macro (frame.cpp:16)


def seafloor_tile(seed, P=2048):
    """Periodic P x P texture, mean ~1, values in [0.15, 2.3]: band-limited noise at three scales + blobs."""
    rng = np.random.default_rng(seed)
    fy = np.fft.fftfreq(P)[:, None]
    fx = np.fft.fftfreq(P)[None, :]
    f2 = fx * fx + fy * fy
    tex = np.zeros((P, P))
    for sigma, amp in ((1.6, 0.55), (3.5, 0.35), (9.0, 0.25)):
        w = rng.standard_normal((P, P))
        g = np.exp(-2.0 * (math.pi ** 2) * (sigma ** 2) * f2)
        s = np.fft.ifft2(np.fft.fft2(w) * g).real
        tex += amp * s / s.std()
    # sparse rocks: bright blob + shadow next to it
    nb = (P * P) // 2500
    ys = rng.integers(0, P, nb); xs = rng.integers(0, P, nb)
    imp = np.zeros((P, P))
    np.add.at(imp, (ys, xs), rng.uniform(0.8, 1.6, nb))
    np.add.at(imp, (ys, (xs + 4) % P), -rng.uniform(0.4, 0.8, nb))
    g = np.exp(-2.0 * (math.pi ** 2) * (1.8 ** 2) * f2)
    blobs = np.fft.ifft2(np.fft.fft2(imp) * g).real
    tex += 6.0 * blobs
    tex = 1.0 + 0.42 * tex / tex.std()
    return np.clip(tex, 0.15, 2.3).astype(np.float32)


class Survey:
    """F legs of N pings x M bins."""

    def __init__(self, F, N, M, seed=20240601, res=0.05, spacing_frac=0.39, drift_xy=0.002, yaw_bias_deg=0.03,
                 noise=0.02, device="cpu", tile=2048, noise_on_device=False):
        self.F, self.N, self.M, self.res, self.seed = F, N, M, res, seed
        # sensor noise from the device's own generator (seeded per frame): the 1000 frames of BASELINE config 5 are 8 M pixels
        # each, and the host generator below would spend a minute on them.  Another noise field than the host one, so a survey
        # made with this switch is a different (equally seeded, equally reproducible) survey.
        self.noise_on_device = noise_on_device
        self.device = torch.device(device)
        rng = np.random.default_rng(seed + 1)
        half = M // 2
        self.gr = (res * np.arange(half)).astype(np.float64)              # ground range per bin
        spacing = spacing_frac * (2 * half * res)
        L = N * res
        poses_true, poses_dr, alts = [], [], []
        drift = np.zeros(2)
        yaw_b = 0.0
        for f in range(F):
            fwd = (f % 2 == 0)
            s = (np.arange(N) + 0.5) * res
            x = s if fwd else (L - s)
            y = np.full(N, f * spacing)
            yaw = np.full(N, 0.0 if fwd else PI_REF)
            # gentle true track wiggle
            y = y + 0.3 * np.sin(2 * math.pi * s / 37.0 + 0.7 * f)
            alt = 10.0 + 1.2 * np.sin(2 * math.pi * s / 83.0 + 0.3 * f) + 0.3 * np.sin(2 * math.pi * s / 17.0 + f)
            z = np.zeros(N)
            truth = np.stack([np.zeros(N), np.zeros(N), yaw, x, y, z], 1)
            # dead-reckoning error: xy random walk + slowly varying yaw bias
            steps = rng.standard_normal((N, 2)) * drift_xy
            walk = drift + np.cumsum(steps, 0)
            drift = walk[-1]
            yaw_b = 0.9 * yaw_b + math.radians(yaw_bias_deg) * rng.standard_normal()
            dr = truth.copy()
            dr[:, 3:5] += walk
            dr[:, 2] += yaw_b
            poses_true.append(truth); poses_dr.append(dr); alts.append(alt)
        self.poses_true = poses_true
        self.poses_dr = poses_dr
        self.alts = alts
        self.noise = noise
        self.tile_np = seafloor_tile(seed, tile)
        self.tile = torch.from_numpy(self.tile_np).to(self.device)
        self.P = tile

    def _sample(self, x, y):
        """bilinear sample of the periodic tile at world (x, y) [torch tensors, float64]"""
        P = self.P
        u = x / self.res; v = y / self.res
        u0 = torch.floor(u); v0 = torch.floor(v)
        fu = (u - u0).to(torch.float32); fv = (v - v0).to(torch.float32)
        iu = torch.remainder(u0.to(torch.int64), P); iv = torch.remainder(v0.to(torch.int64), P)
        iu1 = torch.remainder(iu + 1, P); iv1 = torch.remainder(iv + 1, P)
        T = self.tile
        a = T[iv, iu] * (1 - fu) + T[iv, iu1] * fu
        b = T[iv1, iu] * (1 - fu) + T[iv1, iu1] * fu
        return a * (1 - fv) + b * fv

    def frame(self, f):
        """raw N x M float64 waterfall of leg f rendered from the TRUE poses (torch tensor on self.device)"""
        N, M = self.N, self.M
        half = M // 2
        dev = self.device
        pose = torch.from_numpy(self.poses_true[f]).to(dev)
        gr = torch.from_numpy(self.gr).to(dev)
        col = torch.arange(M, device=dev)
        idx = torch.where(col >= half, col - half, torch.clamp(half - col, max=half - 1))
        g = gr[idx][None, :]
        sign = torch.where(col >= half, 1.0, -1.0).to(torch.float64)[None, :]
        yaw = pose[:, 2:3]
        ang = yaw + sign * (PI_REF / 2)
        x = pose[:, 3:4] + g * torch.cos(ang)
        y = pose[:, 4:5] + g * torch.sin(ang)
        img = self._sample(x, y).to(torch.float64)
        # range-dependent gain ripple + additive sensor noise (seeded per frame)
        if self.noise_on_device and dev.type != "cpu":
            gen = torch.Generator(device=dev); gen.manual_seed(self.seed * 1000 + f)
            nz = torch.randn((N, M), generator=gen, dtype=torch.float32, device=dev).to(torch.float64)
        else:
            gen = torch.Generator(device="cpu"); gen.manual_seed(self.seed * 1000 + f)
            nz = torch.randn((N, M), generator=gen, dtype=torch.float32).to(dev).to(torch.float64)
        gain = 1.0 + 0.05 * torch.cos(g / 7.0)
        img = torch.clamp(img * gain + self.noise * nz, min=0.02) * 1000.0
        return img

    def inputs(self, f):
        """(pose_dr N x 6, altitude N, ground_range M/2) as float64 numpy"""
        return self.poses_dr[f], self.alts[f], self.gr
