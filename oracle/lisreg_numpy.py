"""Independent numpy/scipy mirror of the registration path — TEST INFRASTRUCTURE ONLY.

Purpose: a second, differently-built restatement (LAPACK eigh / SVD least squares / LU solve, scipy cKDTree
neighbours) against which oracle/lisreg_oracle.c is cross-validated, and the generator of tests/golden/*.npz.
Follows /root/reference/src/node/odomEstimationNode.cpp:596-974 (and the label-weighted copies,
subMapOptmizationNode.cpp:1557-1966).  PARITY UNPINNED by the reference (it has no tests).
"""
from __future__ import annotations

import numpy as np
from scipy.spatial import cKDTree

f32 = np.float32


def pose_to_matrix(T):
    """pcl::getTransformation (common.cpp:54-57) in float32; returns 3x4."""
    T = np.asarray(T, f32)
    A, B = np.cos(T[2]), np.sin(T[2])
    Cc, D = np.cos(T[1]), np.sin(T[1])
    E, F = np.cos(T[0]), np.sin(T[0])
    return np.array([[A * Cc, A * D * F - B * E, B * F + A * D * E, T[3]],
                     [B * Cc, A * E + B * D * F, B * D * E - A * F, T[4]],
                     [-D, Cc * F, Cc * E, T[5]]], f32)


def knn5(tgt_xyz, q):
    """exact 5-NN; distances recomputed in float32 from the float32 coordinates (FLANN L2_Simple)."""
    if len(tgt_xyz) < 5:
        return None, None
    tree = cKDTree(tgt_xyz.astype(np.float64))
    _, idx = tree.query(q.astype(np.float64), k=5)
    d = (tgt_xyz[idx] - q[:, None, :]).astype(f32)
    sq = (d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]) + d[..., 2] * d[..., 2]
    order = np.argsort(sq, axis=1, kind="stable")
    return np.take_along_axis(idx, order, 1), np.take_along_axis(sq, order, 1)


def corner_stage(tgt_xyz, src_xyz, w, T, p):
    """cornerOptimization :633-747.  Returns accept flags and coeff[n,4]."""
    M = pose_to_matrix(T)
    q = (src_xyz @ M[:, :3].T + M[:, 3]).astype(f32)
    n = len(src_xyz)
    flags = np.zeros(n, bool); coeff = np.zeros((n, 4), f32)
    idx, sq = knn5(tgt_xyz, q)
    if idx is None:
        return flags, coeff
    ok = sq[:, 4] < p["knn_sq_thresh"]
    nb = tgt_xyz[idx]                                        # n,5,3
    c = nb.sum(1, dtype=f32) / f32(5)
    a = nb - c[:, None, :]
    cov = np.einsum("nji,njk->nik", a, a).astype(f32) / f32(5)
    wv, V = np.linalg.eigh(cov)                              # ascending
    l0, l1 = wv[:, 2], wv[:, 1]
    v0 = V[:, :, 2]
    ok &= l0 > p["line_ratio"] * l1
    p1 = (c.astype(np.float64) + 0.1 * v0.astype(np.float64)).astype(f32)
    p2 = (c.astype(np.float64) - 0.1 * v0.astype(np.float64)).astype(f32)
    x0, y0, z0 = q[:, 0], q[:, 1], q[:, 2]
    x1, y1, z1 = p1.T; x2, y2, z2 = p2.T
    m11 = (x0 - x1) * (y0 - y2) - (x0 - x2) * (y0 - y1)
    m22 = (x0 - x1) * (z0 - z2) - (x0 - x2) * (z0 - z1)
    m33 = (y0 - y1) * (z0 - z2) - (y0 - y2) * (z0 - z1)
    with np.errstate(all="ignore"):
        a012 = np.sqrt(m11 * m11 + m22 * m22 + m33 * m33)
        l12 = np.sqrt((x1 - x2) ** 2 + (y1 - y2) ** 2 + (z1 - z2) ** 2)
        la = ((y1 - y2) * m11 + (z1 - z2) * m22) / a012 / l12
        lb = -((x1 - x2) * m11 - (z1 - z2) * m33) / a012 / l12
        lc = -((x1 - x2) * m22 + (y1 - y2) * m33) / a012 / l12
        ld2 = a012 / l12
        s = (1.0 - 0.9 * np.abs(ld2).astype(np.float64)).astype(f32)
    ws = (w * s).astype(f32)
    coeff = np.stack([ws * la, ws * lb, ws * lc, ws * ld2], 1).astype(f32)
    flags = ok & (s > p["accept_s"])
    coeff[~flags] = 0
    return flags, coeff


def surf_stage(tgt_xyz, src_xyz, w, T, p):
    """surfOptimization :749-827."""
    M = pose_to_matrix(T)
    q = (src_xyz @ M[:, :3].T + M[:, 3]).astype(f32)
    n = len(src_xyz)
    flags = np.zeros(n, bool); coeff = np.zeros((n, 4), f32)
    idx, sq = knn5(tgt_xyz, q)
    if idx is None:
        return flags, coeff
    ok = sq[:, 4] < p["knn_sq_thresh"]
    nb = tgt_xyz[idx].astype(f32)
    # least squares A x = -1 through the SVD pseudo-inverse in float64 (independent of the oracle's float QR)
    X = -(np.linalg.pinv(nb.astype(np.float64)) @ np.ones((5, 1)))[..., 0]
    X = X.astype(f32)
    with np.errstate(all="ignore"):
        ps = np.sqrt((X * X).sum(1, dtype=f32))
        nrm = X / ps[:, None]
        pd = f32(1) / ps
        resid = np.abs(np.einsum("njk,nk->nj", nb, nrm) + pd[:, None])
        ok &= (resid <= p["plane_tol"]).all(1)
        pd2 = (nrm * q).sum(1, dtype=f32) + pd
        rng = np.sqrt(np.sqrt((q * q).sum(1, dtype=f32)))
        s = (1.0 - 0.9 * np.abs(pd2).astype(np.float64) / rng.astype(np.float64)).astype(f32)
    ws = (w * s).astype(f32)
    coeff = np.concatenate([ws[:, None] * nrm, (ws * pd2)[:, None]], 1).astype(f32)
    flags = ok & (s > p["accept_s"])
    coeff[~flags] = 0
    return flags, coeff


def jacobian(T, ori, coeff):
    """LMOptimization rows :862-915 (float32).  ori = UNtransformed source points."""
    T = np.asarray(T, f32)
    srx, crx = np.sin(T[1]), np.cos(T[1])
    sry, cry = np.sin(T[2]), np.cos(T[2])
    srz, crz = np.sin(T[0]), np.cos(T[0])
    px, py, pz = ori[:, 1], ori[:, 2], ori[:, 0]
    cx, cy, cz = coeff[:, 1], coeff[:, 2], coeff[:, 0]
    arx = ((crx * sry * srz * px + crx * crz * sry * py - srx * sry * pz) * cx
           + (-srx * srz * px - crz * srx * py - crx * pz) * cy
           + (crx * cry * srz * px + crx * cry * crz * py - cry * srx * pz) * cz)
    ary = (((cry * srx * srz - crz * sry) * px + (sry * srz + cry * crz * srx) * py + crx * cry * pz) * cx
           + ((-cry * crz - srx * sry * srz) * px + (cry * srz - crz * srx * sry) * py - crx * sry * pz) * cz)
    arz = (((crz * srx * sry - cry * srz) * px + (-cry * crz - srx * sry * srz) * py) * cx
           + (crx * crz * px - crx * srz * py) * cy
           + ((sry * srz + cry * crz * srx) * px + (crz * sry - cry * srx * srz) * py) * cz)
    A = np.stack([arz, arx, ary, cz, cx, cy], 1).astype(f32)
    return A, (-coeff[:, 3]).astype(f32)


def label_weights(labels, p):
    if not p["use_label_weight"]:
        return np.ones(len(labels), f32)
    score = np.asarray(p["label_score"], f32)
    lab = np.asarray(labels).astype(np.int64)
    sc = np.where(lab < 32, score[np.minimum(lab, 31)], 0.0)          # std::map::operator[] on a missing label -> 0 -> w = 2
    return (2.0 - sc.astype(np.float64)).astype(f32)


def align(tgt_c, tgt_s, src_c, src_s, lab_c, lab_s, T_init, p, degenerate_in=0):
    """scan2SubMapOptimization :596-626 without transformUpdate.  Clouds are float32 [n,3].
    Returns (T, stats, trace list of dict)."""
    T = np.array(T_init, f32).copy()
    stats = dict(iters=0, deltaR=100.0, deltaT=100.0, degenerate=int(degenerate_in), n_corr_last=0, status=0)
    if not (len(src_c) > p["edge_min"] and len(src_s) > p["surf_min"]):
        stats["status"] = 1
        return T, stats, []
    wc, ws = label_weights(lab_c, p), label_weights(lab_s, p)
    P = np.zeros((6, 6), f32)
    is_deg = bool(degenerate_in)
    bound = p["fixed_iters"] if p["fixed_iters"] > 0 else p["max_iters"]
    trace, solved, it = [], False, 0
    while it < bound:
        fc, cc = corner_stage(tgt_c, src_c, wc, T, p) if not (p["skip_empty_target"] and len(tgt_c) == 0) \
            else (np.zeros(len(src_c), bool), np.zeros((len(src_c), 4), f32))
        fs, cs = surf_stage(tgt_s, src_s, ws, T, p) if not (p["skip_empty_target"] and len(tgt_s) == 0) \
            else (np.zeros(len(src_s), bool), np.zeros((len(src_s), 4), f32))
        ori = np.concatenate([src_c[fc], src_s[fs]], 0)
        sel = np.concatenate([cc[fc], cs[fs]], 0)
        n_sel = len(ori)
        stats["n_corr_last"] = n_sel
        rec = dict(n_corr=n_sel, solved=False, T=T.copy())
        if n_sel < p["min_corr"]:
            trace.append(rec); it += 1
            continue
        solved = True
        A, B = jacobian(T, ori, sel)
        AtA = (A.astype(np.float64).T @ A.astype(np.float64)).astype(f32)
        AtB = (A.astype(np.float64).T @ B.astype(np.float64)).astype(f32)
        X = np.linalg.solve(AtA.astype(np.float64), AtB.astype(np.float64)).astype(f32)
        if it == 0:
            E, Vc = np.linalg.eigh(AtA.astype(np.float64))
            E = E[::-1]; V = Vc[:, ::-1].T                    # descending, eigenvectors in rows
            V2 = V.copy(); is_deg = False
            for i in range(5, -1, -1):
                if E[i] < p["eig_thresh"]:
                    V2[i] = 0; is_deg = True
                else:
                    break
            P = (np.linalg.inv(V) @ V2).astype(f32)
        elif p["emulate_matp_shadow"]:
            P = np.zeros((6, 6), f32)
        if is_deg:
            X = (P.astype(np.float64) @ X.astype(np.float64)).astype(f32)
        T = (T + X).astype(f32)
        dR = float(np.sqrt(((X[:3] * f32(57.29578)).astype(np.float64) ** 2).sum()))
        dT = float(np.sqrt(((X[3:] * f32(100)).astype(np.float64) ** 2).sum()))
        stats["deltaR"], stats["deltaT"] = dR, dT
        rec.update(solved=True, AtA=AtA, AtB=AtB, X=X, T=T.copy())
        trace.append(rec)
        if (np.float32(dR) < p["conv_deg"] and np.float32(dT) < p["conv_cm"]) and p["fixed_iters"] <= 0:
            break
        it += 1
    stats["iters"] = it
    stats["degenerate"] = int(is_deg)
    stats["status"] = 0 if solved else 2
    return T, stats, trace


def extract_features(x, y, z, ring, p):
    """Independent Python mirror of LaserProcessing's projection + feature extraction
    (/root/reference/src/core/laserProcessing.cpp:467-713, no IMU de-skew) for SMALL scans — plain loops, written
    from the reference text separately from oracle/lisreg_oracle.c.  p: dict(n_scan, horizon_scan, downsample_rate,
    min_range, max_range, edge_threshold, surf_threshold).  Returns index lists into the input arrays, with the same
    defined behaviour as the oracle where the reference has none (zero-initialised per-frame arrays, bounds-checked
    +-5 accesses, (value, index) order for equal curvatures)."""
    H, W = p["n_scan"], p["horizon_scan"]
    owner = -np.ones((H, W), np.int64)
    rmat = np.full((H, W), np.inf, np.float32)
    ang_res = f32(360.0 / float(f32(W)))
    for i in range(len(x)):
        rng_ = f32(np.sqrt(f32(f32(x[i] * x[i] + y[i] * y[i]) + z[i] * z[i])))
        if rng_ < p["min_range"] or rng_ > p["max_range"]:
            continue
        row = int(ring[i])
        if row < 0 or row >= H or row % p["downsample_rate"] != 0:
            continue
        ha = f32(float(f32(f32(np.arctan2(float(x[i]), float(y[i]))) * f32(180))) / np.pi)   # double atan2 rounded to float
        col = int(-np.round((float(ha) - 90.0) / float(ang_res)) + W // 2)
        if col >= W:
            col -= W
        if col < 0 or col >= W or owner[row, col] >= 0:
            continue
        owner[row, col] = i
        rmat[row, col] = rng_
    src, col_ind, pr, start, end = [], [], [], [], []
    for i in range(H):
        start.append(len(src) - 1 + 5)
        for j in range(W):
            if owner[i, j] >= 0:
                src.append(int(owner[i, j])); col_ind.append(j); pr.append(rmat[i, j])
        end.append(len(src) - 1 - 5)
    n = len(src)
    pr = np.array(pr + [0.0] * 16, f32)
    curv = np.zeros(n + 16, f32); picked = np.zeros(n + 16, np.int64); label = np.zeros(n + 16, np.int64)
    for i in range(5, n - 5):
        d = f32(0)
        for t in (pr[i - 5], pr[i - 4], pr[i - 3], pr[i - 2], pr[i - 1]):
            d = f32(d + t)
        d = f32(d - f32(pr[i] * f32(10)))
        for t in (pr[i + 1], pr[i + 2], pr[i + 3], pr[i + 4], pr[i + 5]):
            d = f32(d + t)
        curv[i] = f32(d * d)
    for i in range(5, n - 6):
        if abs(col_ind[i + 1] - col_ind[i]) < 10:
            if float(f32(pr[i] - pr[i + 1])) > 0.3:
                picked[i - 5:i + 1] = 1
            elif float(f32(pr[i + 1] - pr[i])) > 0.3:
                picked[i + 1:i + 7] = 1
        if float(abs(f32(pr[i - 1] - pr[i]))) > 0.02 * float(pr[i]) and float(abs(f32(pr[i + 1] - pr[i]))) > 0.02 * float(pr[i]):
            picked[i] = 1

    def suppress(ind):
        for l in range(1, 6):
            if ind + l >= n or ind + l - 1 < 0 or abs(col_ind[ind + l] - col_ind[ind + l - 1]) > 10:
                break
            picked[ind + l] = 1
        for l in range(-1, -6, -1):
            if ind + l < 0 or ind + l + 1 >= n or abs(col_ind[ind + l] - col_ind[ind + l + 1]) > 10:
                break
            picked[ind + l] = 1

    corner, surface, csharp, ssharp = [], [], [], []
    for i in range(H):
        for j in range(6):
            sp = (start[i] * (6 - j) + end[i] * j) // 6 if (start[i] * (6 - j) + end[i] * j) >= 0 else -((-(start[i] * (6 - j) + end[i] * j)) // 6)
            e_ = start[i] * (5 - j) + end[i] * (j + 1)
            ep = (e_ // 6 if e_ >= 0 else -((-e_) // 6)) - 1          # C integer division truncates toward zero
            if sp >= ep:
                continue
            order = sorted(range(sp, ep), key=lambda k: (float(curv[k]), k)) + [ep]
            cnt = 0
            for k in range(ep, sp - 1, -1):
                ind = order[k - sp]
                if picked[ind] == 0 and curv[ind] > p["edge_threshold"]:
                    cnt += 1
                    if cnt <= 20:
                        label[ind] = 1; corner.append(src[ind])
                        if cnt <= 4:
                            csharp.append(src[ind])
                    else:
                        break
                    picked[ind] = 1; suppress(ind)
            cnt = 0
            for k in range(sp, ep + 1):
                ind = order[k - sp]
                if picked[ind] == 0 and curv[ind] < p["surf_threshold"]:
                    cnt += 1; label[ind] = -1; picked[ind] = 1
                    if cnt <= 10:
                        ssharp.append(src[ind])
                    suppress(ind)
            for k in range(sp, ep + 1):
                if label[k] <= 0:
                    surface.append(src[k])
    return dict(deskewed=np.array(src, np.int32), corner=np.array(corner, np.int32), surface=np.array(surface, np.int32),
                corner_sharp=np.array(csharp, np.int32), surface_sharp=np.array(ssharp, np.int32))
