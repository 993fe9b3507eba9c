"""Seeded synthetic KITTI-shape scenes for parity tests and bench.py (SURVEY.md §8d "Synthetic scene generator").

World: box room 80 x 80 m (4 walls, floor z=0) + 64 vertical poles (r=0.15 m, h=6 m) on a jittered 8x8 grid.
Scan : H x W rays, elevations linear in [-24.8, +2.0] deg (HDL-64 span, laserPretreatmentNode.cpp:113-117),
       azimuth 360/W, nearest hit, Gaussian range noise sigma=0.02 m, max range 70 m (config/params.yaml:74).
       pole hit -> corner feature (label 18), floor -> planar (label 9), wall -> planar (label 13).
Submap: M points sampled area-uniformly from the same surfaces (+ sigma 0.02 m), ~5 % on poles.
Clouds are returned in the reference's host layout: PCL 32-byte point structs (common.h:9,25-35).
"""
from __future__ import annotations

import numpy as np

ROOM_HALF = 40.0
POLE_R = 0.15
POLE_H = 6.0
SENSOR_Z = 1.73
MAX_RANGE = 70.0
LABEL_POLE, LABEL_GROUND, LABEL_WALL = 18, 9, 13

PCL_DTYPE = np.dtype({
    "names": ["x", "y", "z", "intensity", "label"],
    "formats": ["<f4", "<f4", "<f4", "<f4", "<u2"],
    "offsets": [0, 4, 8, 16, 20],
    "itemsize": 32,
})


def pole_centers(scene_seed: int = 1234) -> np.ndarray:
    rng = np.random.default_rng(scene_seed)
    g = (np.arange(8) + 0.5) * (2 * ROOM_HALF / 8) - ROOM_HALF
    cx, cy = np.meshgrid(g, g, indexing="ij")
    c = np.stack([cx.ravel(), cy.ravel()], 1) + rng.uniform(-2.0, 2.0, (64, 2))
    return c.astype(np.float64)


def to_pcl(xyz: np.ndarray, label=None, intensity=None) -> np.ndarray:
    """Pack xyz[n,3] (+ labels) into an array of 32-byte PCL PointXYZI / PointXYZIL structs."""
    n = xyz.shape[0]
    out = np.zeros(n, dtype=PCL_DTYPE)
    out["x"], out["y"], out["z"] = xyz[:, 0], xyz[:, 1], xyz[:, 2]
    if intensity is not None:
        out["intensity"] = intensity
    if label is not None:
        out["label"] = label
    return out


def concat_clouds(clouds, dtype=None) -> np.ndarray:
    """`*a += *b` for PCL struct arrays.  np.concatenate would re-pack a padded struct dtype (32-byte PCL points become 18-byte
    records with the label at another offset); this keeps the layout of the first cloud."""
    clouds = list(clouds)
    dtype = dtype or clouds[0].dtype
    out = np.zeros(sum(len(c) for c in clouds), dtype)
    o = 0
    for c in clouds:
        assert c.dtype == dtype, (c.dtype, dtype)
        out[o:o + len(c)] = c
        o += len(c)
    return out


def pcl_xyz(cloud: np.ndarray) -> np.ndarray:
    return np.stack([cloud["x"], cloud["y"], cloud["z"]], 1).astype(np.float32)


def pose_matrix(T) -> np.ndarray:
    """float64 4x4 of pose T = [roll, pitch, yaw, x, y, z] (Rz*Ry*Rx, common.cpp:54-57)."""
    r, p, y = float(T[0]), float(T[1]), float(T[2])
    A, B, C, D, E, F = np.cos(y), np.sin(y), np.cos(p), np.sin(p), np.cos(r), np.sin(r)
    M = np.eye(4)
    M[:3, :3] = [[A * C, A * D * F - B * E, B * F + A * D * E],
                 [B * C, A * E + B * D * F, B * D * E - A * F],
                 [-D, C * F, C * E]]
    M[:3, 3] = T[3:6]
    return M


def make_submap(m_points: int, seed: int = 42, scene_seed: int = 1234, labelled: bool = False):
    """Returns (corner_cloud, surf_cloud) as PCL struct arrays; ~5 % of points on poles."""
    rng = np.random.default_rng(seed)
    m_c = int(round(0.05 * m_points))
    m_p = m_points - m_c
    poles = pole_centers(scene_seed)
    k = rng.integers(0, 64, m_c)
    ang = rng.uniform(0, 2 * np.pi, m_c)
    pc = np.stack([poles[k, 0] + POLE_R * np.cos(ang), poles[k, 1] + POLE_R * np.sin(ang),
                   rng.uniform(0, POLE_H, m_c)], 1)
    a_floor = (2 * ROOM_HALF) ** 2
    wall_h = 12.0
    a_wall = 4 * (2 * ROOM_HALF) * wall_h
    n_floor = int(round(m_p * a_floor / (a_floor + a_wall)))
    n_wall = m_p - n_floor
    fl = np.stack([rng.uniform(-ROOM_HALF, ROOM_HALF, n_floor), rng.uniform(-ROOM_HALF, ROOM_HALF, n_floor),
                   np.zeros(n_floor)], 1)
    side = rng.integers(0, 4, n_wall)
    u = rng.uniform(-ROOM_HALF, ROOM_HALF, n_wall)
    h = rng.uniform(0, wall_h, n_wall)
    wx = np.where(side == 0, ROOM_HALF, np.where(side == 1, -ROOM_HALF, u))
    wy = np.where(side == 2, ROOM_HALF, np.where(side == 3, -ROOM_HALF, u))
    wl = np.stack([wx, wy, h], 1)
    ps = np.concatenate([fl, wl], 0)
    pc = pc + rng.normal(0, 0.02, pc.shape)
    ps = ps + rng.normal(0, 0.02, ps.shape)
    lab_c = np.full(m_c, LABEL_POLE if labelled else 0, np.uint16)
    lab_s = np.concatenate([np.full(n_floor, LABEL_GROUND if labelled else 0, np.uint16),
                            np.full(n_wall, LABEL_WALL if labelled else 0, np.uint16)])
    perm = rng.permutation(m_p)
    return to_pcl(pc.astype(np.float32), lab_c), to_pcl(ps[perm].astype(np.float32), lab_s[perm])


def draw_pose(rng) -> np.ndarray:
    """Ground-truth sensor pose inside the room: level, random yaw, position in the central 40x40 m."""
    return np.array([0.0, 0.0, rng.uniform(-np.pi, np.pi), rng.uniform(-20, 20), rng.uniform(-20, 20), SENSOR_Z],
                    np.float64)


def perturb_pose(T_true: np.ndarray, rng, trans=0.3, rot_deg=2.0) -> np.ndarray:
    d = np.concatenate([np.deg2rad(rng.uniform(-rot_deg, rot_deg, 3)), rng.uniform(-trans, trans, 3)])
    return (T_true + d).astype(np.float32)


def make_scan(h: int, w: int, seed: int, scene_seed: int = 1234, labelled: bool = False, T_true=None):
    """Ray-cast one H x W scan.  Returns dict(corner=PCL array, surf=PCL array, T_true=float64[6]);
    points are in the SENSOR frame, ordered ring-major (row-major range image), invalid pixels dropped."""
    rng = np.random.default_rng(seed)
    if T_true is None:
        T_true = draw_pose(rng)
    M = pose_matrix(T_true)
    el = np.deg2rad(np.linspace(-24.8, 2.0, h))
    az = np.deg2rad(np.arange(w) * (360.0 / w))
    ce, se = np.cos(el)[:, None], np.sin(el)[:, None]
    d_s = np.stack([ce * np.cos(az)[None, :], ce * np.sin(az)[None, :], np.broadcast_to(se, (h, w))], -1).reshape(-1, 3)
    d_w = d_s @ M[:3, :3].T
    o = M[:3, 3]
    n = d_w.shape[0]
    best = np.full(n, np.inf)
    kind = np.zeros(n, np.int8)          # 0 none, 1 floor, 2 wall, 3 pole
    with np.errstate(divide="ignore", invalid="ignore"):
        t = -o[2] / d_w[:, 2]
        hit = (d_w[:, 2] < 0) & (t > 0)
        px, py = o[0] + t * d_w[:, 0], o[1] + t * d_w[:, 1]
        hit &= (np.abs(px) <= ROOM_HALF) & (np.abs(py) <= ROOM_HALF)
        best = np.where(hit, t, best); kind = np.where(hit, 1, kind)
        for axis in (0, 1):
            for sgn in (+1.0, -1.0):
                t = (sgn * ROOM_HALF - o[axis]) / d_w[:, axis]
                other = o[1 - axis] + t * d_w[:, 1 - axis]
                z = o[2] + t * d_w[:, 2]
                hit = (t > 0) & (np.abs(other) <= ROOM_HALF) & (z >= 0) & (z <= 12.0) & (t < best)
                best = np.where(hit, t, best); kind = np.where(hit, 2, kind)
        poles = pole_centers(scene_seed)
        a = d_w[:, 0] ** 2 + d_w[:, 1] ** 2
        for cx, cy in poles:
            ox, oy = o[0] - cx, o[1] - cy
            b = ox * d_w[:, 0] + oy * d_w[:, 1]
            c = ox * ox + oy * oy - POLE_R ** 2
            disc = b * b - a * c
            t = (-b - np.sqrt(np.maximum(disc, 0))) / a
            z = o[2] + t * d_w[:, 2]
            hit = (disc > 0) & (t > 0) & (z >= 0) & (z <= POLE_H) & (t < best)
            best = np.where(hit, t, best); kind = np.where(hit, 3, kind)
    rngs = best + rng.normal(0, 0.02, n)
    valid = (kind > 0) & (rngs < MAX_RANGE) & (rngs > 0.5)
    pts = (d_s * np.where(valid, rngs, 0.0)[:, None]).astype(np.float32)
    lab = np.select([kind == 3, kind == 1, kind == 2], [LABEL_POLE, LABEL_GROUND, LABEL_WALL], 0).astype(np.uint16)
    if not labelled:
        lab = np.zeros_like(lab)
    is_c = valid & (kind == 3)
    is_s = valid & (kind != 3)
    return dict(corner=to_pcl(pts[is_c], lab[is_c]), surf=to_pcl(pts[is_s], lab[is_s]), T_true=T_true)


def make_case(h=16, w=450, m_points=20000, scan_seed=1000, submap_seed=42, labelled=False, trans=0.3, rot_deg=2.0,
              local_radius=None, pose_xy=None):
    """One registration problem: target clouds, source clouds, initial guess, ground truth.
    local_radius: keep only the part of the scene within that distance of the sensor (small fixtures at full
    point density: generate a big submap, keep the neighbourhood)."""
    tc, ts = make_submap(m_points, submap_seed, labelled=labelled)
    T_fix = None
    if pose_xy is not None:                      # put the sensor somewhere specific (e.g. near a room corner)
        T_fix = draw_pose(np.random.default_rng(scan_seed))
        T_fix[3], T_fix[4] = pose_xy
    sc = make_scan(h, w, scan_seed, labelled=labelled, T_true=T_fix)
    rng = np.random.default_rng(scan_seed + 7919)
    T0 = perturb_pose(sc["T_true"], rng, trans, rot_deg)
    src_c, src_s = sc["corner"], sc["surf"]
    if local_radius is not None:
        cx, cy = sc["T_true"][3], sc["T_true"][4]
        tc = tc[np.hypot(tc["x"] - cx, tc["y"] - cy) < local_radius + 1.5]
        ts = ts[np.hypot(ts["x"] - cx, ts["y"] - cy) < local_radius + 1.5]
        src_c = src_c[np.sqrt(src_c["x"] ** 2 + src_c["y"] ** 2 + src_c["z"] ** 2) < local_radius]
        src_s = src_s[np.sqrt(src_s["x"] ** 2 + src_s["y"] ** 2 + src_s["z"] ** 2) < local_radius]
    return dict(tgt_corner=tc, tgt_surf=ts, src_corner=src_c, src_surf=src_s,
                T_init=T0, T_true=sc["T_true"].astype(np.float32))


def make_plane_case(n_tgt=6000, n_src=1500, seed=5, half=8.0, trans=0.2, rot_deg=1.0):
    """Degenerate scene: one horizontal plane only (x, y, yaw unobservable) — pins the iteration-0 degeneracy
    projector and the local-matP quirk of LMOptimization (odomEstimationNode.cpp:923-953)."""
    rng = np.random.default_rng(seed)
    tgt = np.stack([rng.uniform(-half, half, n_tgt), rng.uniform(-half, half, n_tgt), rng.normal(0, 0.01, n_tgt)], 1)
    T_true = np.array([0.0, 0.0, 0.3, 0.5, -0.4, 1.5])
    M = pose_matrix(T_true)
    world = np.stack([rng.uniform(-half + 2, half - 2, n_src), rng.uniform(-half + 2, half - 2, n_src),
                      rng.normal(0, 0.01, n_src)], 1)
    src = (world - M[:3, 3]) @ M[:3, :3]            # sensor frame: R^T (p - t)
    T0 = perturb_pose(T_true, rng, trans, rot_deg)
    empty = to_pcl(np.zeros((0, 3), np.float32), np.zeros(0, np.uint16))
    return dict(tgt_corner=empty, tgt_surf=to_pcl(tgt.astype(np.float32), np.zeros(n_tgt, np.uint16)),
                src_corner=empty, src_surf=to_pcl(src.astype(np.float32), np.zeros(n_src, np.uint16)),
                T_init=T0, T_true=T_true.astype(np.float32))


# PointXYZIRT (src/include/common.h:12-23): x y z pad intensity, uint16 ring @20, float time @24; 32 bytes
XYZIRT_DTYPE = np.dtype({"names": ["x", "y", "z", "intensity", "ring", "time"],
                         "formats": ["<f4", "<f4", "<f4", "<f4", "<u2", "<f4"],
                         "offsets": [0, 4, 8, 16, 20, 24], "itemsize": 32})


def make_raw_scan(h: int, w: int, seed: int, scene_seed: int = 1234, dup_fraction: float = 0.05, shuffle: bool = False):
    """A raw LiDAR sweep as laserPretreatment hands it to LaserProcessing (ring + time channels, valid returns only),
    plus a few jittered duplicates so that several points compete for one range-image pixel (first one must win)."""
    rng = np.random.default_rng(seed + 31)
    sc = make_scan(h, w, seed, scene_seed)
    both = concat_clouds([sc["corner"], sc["surf"]])
    xyz = pcl_xyz(both)
    el = np.degrees(np.arctan2(xyz[:, 2], np.hypot(xyz[:, 0], xyz[:, 1])))
    ring = np.clip(np.rint((el + 24.8) / (26.8 / (h - 1))), 0, h - 1).astype(np.uint16)
    order = np.lexsort((np.arctan2(xyz[:, 1], xyz[:, 0]), ring))          # a sweep: ring-major, azimuth ascending
    xyz, ring = xyz[order], ring[order]
    k = int(dup_fraction * len(xyz))
    pick = rng.integers(0, len(xyz), k)
    xyz = np.concatenate([xyz, xyz[pick] * (1 + rng.normal(0, 1e-3, (k, 1))).astype(np.float32)])
    ring = np.concatenate([ring, ring[pick]])
    if shuffle:
        perm = rng.permutation(len(xyz)); xyz, ring = xyz[perm], ring[perm]
    out = np.zeros(len(xyz), XYZIRT_DTYPE)
    out["x"], out["y"], out["z"] = xyz[:, 0], xyz[:, 1], xyz[:, 2]
    out["intensity"] = rng.uniform(0, 255, len(xyz)).astype(np.float32)
    out["ring"] = ring
    out["time"] = np.linspace(0, 0.1, len(xyz), dtype=np.float32)
    return out
