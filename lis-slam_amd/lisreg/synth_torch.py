"""Device-side (torch, plumbing only) generator of the SAME synthetic scene as synth.py, for bench.py: ray-casts
KITTI-shape scans directly into HBM as 16-byte lisreg_dpoint records so the timed region starts with inputs
resident on the GPU.  Geometry constants and the pole layout come from synth.py (seeded numpy)."""
from __future__ import annotations

import math

import numpy as np
import torch

from . import synth


def _pose_matrix_t(T, device):
    return torch.tensor(synth.pose_matrix(T), dtype=torch.float64, device=device)


def make_scan_device(h: int, w: int, seed: int, device, scene_seed: int = 1234, labelled: bool = False):
    """Returns (corner[n_c,4] float32, surf[n_s,4] float32, T_true float64[6]) — device tensors of lisreg_dpoint."""
    rng = np.random.default_rng(seed)
    T_true = synth.draw_pose(rng)
    M = _pose_matrix_t(T_true, device)
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    el = torch.deg2rad(torch.linspace(-24.8, 2.0, h, dtype=torch.float64, device=device))
    az = torch.deg2rad(torch.arange(w, dtype=torch.float64, device=device) * (360.0 / w))
    ce, se = torch.cos(el)[:, None], torch.sin(el)[:, None]
    d_s = torch.stack([ce * torch.cos(az)[None, :], ce * torch.sin(az)[None, :], se.expand(h, w)], -1).reshape(-1, 3)
    d_w = d_s @ M[:3, :3].T
    o = M[:3, 3]
    n = d_w.shape[0]
    inf = float("inf")
    best = torch.full((n,), inf, dtype=torch.float64, device=device)
    kind = torch.zeros(n, dtype=torch.int8, device=device)
    R = synth.ROOM_HALF
    t = -o[2] / d_w[:, 2]
    px, py = o[0] + t * d_w[:, 0], o[1] + t * d_w[:, 1]
    hit = (d_w[:, 2] < 0) & (t > 0) & (px.abs() <= R) & (py.abs() <= R)
    best = torch.where(hit, t, best); kind = torch.where(hit, torch.ones_like(kind), kind)
    for axis in (0, 1):
        for sgn in (1.0, -1.0):
            t = (sgn * R - o[axis]) / d_w[:, axis]
            other = o[1 - axis] + t * d_w[:, 1 - axis]
            z = o[2] + t * d_w[:, 2]
            hit = (t > 0) & (other.abs() <= R) & (z >= 0) & (z <= 12.0) & (t < best)
            best = torch.where(hit, t, best); kind = torch.where(hit, torch.full_like(kind, 2), kind)
    poles = torch.tensor(synth.pole_centers(scene_seed), dtype=torch.float64, device=device)   # [64,2]
    a = d_w[:, 0] ** 2 + d_w[:, 1] ** 2                                                        # [n]
    ox, oy = (o[0] - poles[:, 0])[None, :], (o[1] - poles[:, 1])[None, :]                      # [1,64]
    b = ox * d_w[:, 0:1] + oy * d_w[:, 1:2]                                                    # [n,64]
    c = ox * ox + oy * oy - synth.POLE_R ** 2
    disc = b * b - a[:, None] * c
    tp = (-b - torch.sqrt(disc.clamp_min(0))) / a[:, None]
    zp = o[2] + tp * d_w[:, 2:3]
    ok = (disc > 0) & (tp > 0) & (zp >= 0) & (zp <= synth.POLE_H)
    tp = torch.where(ok, tp, torch.full_like(tp, inf)).min(dim=1).values
    hit = tp < best
    best = torch.where(hit, tp, best); kind = torch.where(hit, torch.full_like(kind, 3), kind)
    rngs = best + 0.02 * torch.randn(n, dtype=torch.float64, device=device, generator=g)
    valid = (kind > 0) & (rngs < synth.MAX_RANGE) & (rngs > 0.5)
    pts = (d_s * torch.where(valid, rngs, torch.zeros_like(rngs))[:, None]).to(torch.float32)
    lab = torch.zeros(n, dtype=torch.int32, device=device)
    if labelled:
        lab = torch.where(kind == 3, synth.LABEL_POLE, torch.where(kind == 1, synth.LABEL_GROUND, synth.LABEL_WALL)).to(torch.int32)
    rec = torch.cat([pts, lab.view(torch.float32)[:, None]], 1).contiguous()
    is_c = valid & (kind == 3)
    is_s = valid & (kind != 3)
    return rec[is_c].contiguous(), rec[is_s].contiguous(), T_true


def submap_device(m_points: int, device, seed: int = 42, labelled: bool = False):
    """The numpy submap of synth.make_submap uploaded as two [n,4] float32 record tensors."""
    from . import pack_device_records
    tc, ts = synth.make_submap(m_points, seed, labelled=labelled)
    return (torch.from_numpy(pack_device_records(tc)).to(device), torch.from_numpy(pack_device_records(ts)).to(device),
            tc, ts)


def records_to_pcl(rec: torch.Tensor) -> np.ndarray:
    """[n,4] device records -> host PCL struct array (for the CPU baseline / parity legs)."""
    h = rec.detach().cpu().numpy()
    return synth.to_pcl(h[:, :3].copy(), h[:, 3].copy().view(np.uint32).astype(np.uint16))
