"""CPU tests of the oracle (oracle/lisreg_oracle.c) against independent library routines and hand-checkable
geometry — the pinning available for a reference that ships no tests of its own."""
import ctypes as C

import numpy as np
import pytest
from scipy.spatial import cKDTree

f32 = np.float32


def fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int))


def test_pose_to_matrix_is_rz_ry_rx(oracle):
    L = oracle.lib()
    rng = np.random.default_rng(0)
    for _ in range(20):
        T = rng.uniform(-1, 1, 6).astype(f32)
        M = np.zeros(12, f32)
        L.orc_pose_to_matrix(fp(T), fp(M))
        r, p, y = T[:3].astype(np.float64)
        Rx = np.array([[1, 0, 0], [0, np.cos(r), -np.sin(r)], [0, np.sin(r), np.cos(r)]])
        Ry = np.array([[np.cos(p), 0, np.sin(p)], [0, 1, 0], [-np.sin(p), 0, np.cos(p)]])
        Rz = np.array([[np.cos(y), -np.sin(y), 0], [np.sin(y), np.cos(y), 0], [0, 0, 1]])
        M = M.reshape(3, 4)
        assert np.allclose(M[:, :3], Rz @ Ry @ Rx, atol=2e-7)
        assert np.array_equal(M[:, 3], T[3:])


@pytest.mark.parametrize("n", [0, 3, 5, 16, 1000, 20000])
def test_kdtree_knn_exact(oracle, n):
    L = oracle.lib()
    rng = np.random.default_rng(n)
    pts = rng.uniform(-20, 20, (n, 3)).astype(f32)
    tree = L.orc_kdtree_build(fp(pts), n, 15)
    ref = cKDTree(pts.astype(np.float64)) if n else None
    for _ in range(200):
        q = rng.uniform(-22, 22, 3).astype(f32)
        idx, sq = np.zeros(5, np.int32), np.zeros(5, f32)
        idx2, sq2 = np.zeros(5, np.int32), np.zeros(5, f32)
        k = L.orc_kdtree_knn(tree, fp(q), 5, ip(idx), fp(sq))
        k2 = L.orc_bruteforce_knn(fp(pts), n, fp(q), 5, ip(idx2), fp(sq2))
        assert k == k2 == min(5, n)
        assert np.array_equal(sq[:k], sq2[:k]) and np.all(np.diff(sq[:k]) >= 0)
        if n >= 5:
            _, ridx = ref.query(q.astype(np.float64), k=5)
            assert set(idx.tolist()) == set(ridx.tolist())
    L.orc_kdtree_free(tree)


def test_knn_threshold_equals_radius_search(oracle):
    """`nearestKSearch(5)` + `sqDist[4] < tau` == 'at least five points strictly inside radius sqrt(tau)'
    (the equivalence that makes a fixed-radius grid search exact; SURVEY.md quick facts)."""
    L = oracle.lib()
    rng = np.random.default_rng(3)
    pts = rng.uniform(-5, 5, (3000, 3)).astype(f32)
    tree = L.orc_kdtree_build(fp(pts), len(pts), 15)
    for tau in (1.0, 2.0, 0.3):
        for _ in range(300):
            q = rng.uniform(-6, 6, 3).astype(f32)
            idx, sq = np.zeros(5, np.int32), np.zeros(5, f32)
            L.orc_kdtree_knn(tree, fp(q), 5, ip(idx), fp(sq))
            d = pts - q
            d2 = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]
            assert (sq[4] < tau) == (np.count_nonzero(d2 < f32(tau)) >= 5)
    L.orc_kdtree_free(tree)


@pytest.mark.parametrize("n", [3, 6])
def test_eigen_sym_matches_lapack(oracle, n):
    L = oracle.lib()
    rng = np.random.default_rng(n)
    for _ in range(50):
        B = rng.normal(size=(n, n))
        A = (B @ B.T * rng.uniform(0.1, 100)).astype(f32)
        W, V = np.zeros(n, f32), np.zeros(n * n, f32)
        L.orc_eigen_sym(fp(np.ascontiguousarray(A)), n, fp(W), fp(V))
        V = V.reshape(n, n)
        w_ref = np.linalg.eigvalsh(A.astype(np.float64))[::-1]
        assert np.all(np.diff(W) <= 0)                                  # descending (OpenCV convention)
        assert np.allclose(W, w_ref, rtol=2e-5, atol=2e-5 * abs(w_ref).max())
        assert np.allclose(V @ V.T, np.eye(n), atol=5e-6)               # rows are orthonormal eigenvectors
        assert np.allclose(A.astype(np.float64) @ V.T.astype(np.float64), V.T * W, atol=3e-5 * abs(w_ref).max())


def test_lstsq5x3_matches_lapack(oracle):
    L = oracle.lib()
    rng = np.random.default_rng(5)
    for _ in range(100):
        n = rng.normal(size=3); n /= np.linalg.norm(n)
        c = rng.uniform(-30, 30, 3)
        basis = np.linalg.svd(n[None, :])[2][1:]
        P = c + rng.uniform(-0.5, 0.5, (5, 2)) @ basis + rng.normal(0, 0.01, (5, 1)) * n
        A = P.astype(f32)
        b = -np.ones(5, f32)
        x = np.zeros(3, f32)
        L.orc_lstsq5x3(fp(np.ascontiguousarray(A)), fp(b), fp(x))
        ref = np.linalg.lstsq(A.astype(np.float64), b.astype(np.float64), rcond=None)[0]
        # QR is backward stable: error ~ eps * cond(A); cond here is |c| / spread ~ 1e2
        assert np.allclose(x, ref, rtol=0, atol=3e-4 * np.abs(ref).max() + 1e-7)


def test_solve6_and_inv6(oracle):
    L = oracle.lib()
    rng = np.random.default_rng(6)
    for _ in range(50):
        B = rng.normal(size=(6, 6))
        A = (B @ B.T + np.eye(6)).astype(f32)
        b = rng.normal(size=6).astype(f32)
        x, Ai = np.zeros(6, f32), np.zeros(36, f32)
        assert L.orc_solve6(fp(np.ascontiguousarray(A)), fp(b), fp(x)) == 1
        assert L.orc_inv6(fp(np.ascontiguousarray(A)), fp(Ai)) == 1
        cond = np.linalg.cond(A.astype(np.float64))
        assert np.allclose(x, np.linalg.solve(A.astype(np.float64), b), rtol=0, atol=3e-6 * cond * np.abs(x).max())
        assert np.allclose(Ai.reshape(6, 6) @ A, np.eye(6), atol=1e-5 * cond)
    Z = np.zeros(36, f32)
    x = np.ones(6, f32)
    assert L.orc_solve6(fp(Z), fp(np.ones(6, f32)), fp(x)) == 0 and not x.any()


def test_corner_coeff_known_line(oracle):
    """Five points on the vertical line x=2,y=3; query at distance 0.25 -> unit gradient toward +x, ld2 = 0.25."""
    L = oracle.lib()
    p = oracle.default_params(1)
    nb = np.array([[2, 3, z] for z in (0.0, 0.2, 0.4, 0.6, 0.8)], f32)
    q = np.array([2.25, 3.0, 0.4], f32)
    cf = np.zeros(4, f32)
    ok = L.orc_corner_coeff(fp(nb), fp(q), 1.0, C.byref(p), fp(cf))
    s = 1 - 0.9 * 0.25
    assert ok == 1
    assert np.allclose(cf, [s * 1, 0, 0, s * 0.25], atol=1e-5)
    # isotropic blob: lambda0 > 3*lambda1 fails
    rng = np.random.default_rng(1)
    blob = (rng.normal(0, 0.1, (5, 3)) + [2, 3, 0.4]).astype(f32)
    blob[:3] = [[2.1, 3, 0.4], [2, 3.1, 0.4], [2, 3, 0.5]]
    assert L.orc_corner_coeff(fp(blob), fp(q), 1.0, C.byref(p), fp(cf)) == 0
    # far from the line: s = 1 - 0.9*d <= 0.1  ->  rejected (d >= 1)
    q_far = np.array([3.05, 3.0, 0.4], f32)
    assert L.orc_corner_coeff(fp(nb), fp(q_far), 1.0, C.byref(p), fp(cf)) == 0


def test_surf_coeff_known_plane(oracle):
    """Five points on z = 1.5; query 0.1 above: normal (0,0,-1) or (0,0,1) with matching sign of pd2."""
    L = oracle.lib()
    p = oracle.default_params(1)
    nb = np.array([[10, 5, 1.5], [10.3, 5, 1.5], [10, 5.3, 1.5], [10.3, 5.3, 1.5], [10.15, 5.1, 1.5]], f32)
    q = np.array([10.1, 5.1, 1.6], f32)
    cf = np.zeros(4, f32)
    assert L.orc_surf_coeff(fp(nb), fp(q), 1.0, C.byref(p), fp(cf)) == 1
    rng_ = np.sqrt(np.sqrt(float((q.astype(np.float64) ** 2).sum())))
    s = 1 - 0.9 * 0.1 / rng_
    # A n = -1 with all z = 1.5 -> n = (0,0,-1/1.5): unit normal (0,0,-1), pd = 1/|n| = 1.5, pd2 = -1.6 + 1.5 = -0.1
    assert np.allclose(cf, [0, 0, -s, -0.1 * s], atol=2e-5)
    # label weight multiplies all four, acceptance still uses s
    assert L.orc_surf_coeff(fp(nb), fp(q), 1.5, C.byref(p), fp(cf)) == 1
    assert np.allclose(cf, [0, 0, -1.5 * s, -0.15 * s], atol=3e-5)
    # one neighbour 0.3 off the plane: |n.p + d| > 0.2 invalidates
    nb2 = nb.copy(); nb2[4, 2] = 2.2
    assert L.orc_surf_coeff(fp(nb2), fp(q), 1.0, C.byref(p), fp(cf)) == 0


def _plane_closed(nb, dtype):
    """plane5_closed of lisreg_assoc.hip (the production build's plane fit), operation for operation, in `dtype`"""
    nb = nb.astype(dtype)
    c = ((nb[0] + nb[1]) + (nb[2] + nb[3]) + nb[4]) * dtype(0.2)
    d = nb - c
    sxx, sxy, sxz = (d[:, 0] * d[:, 0]).sum(dtype=dtype), (d[:, 0] * d[:, 1]).sum(dtype=dtype), (d[:, 0] * d[:, 2]).sum(dtype=dtype)
    syy, syz, szz = (d[:, 1] * d[:, 1]).sum(dtype=dtype), (d[:, 1] * d[:, 2]).sum(dtype=dtype), (d[:, 2] * d[:, 2]).sum(dtype=dtype)
    axx, axy, axz = syy * szz - syz * syz, sxz * syz - sxy * szz, sxy * syz - sxz * syy
    ayy, ayz, azz = sxx * szz - sxz * sxz, sxy * sxz - sxx * syz, sxx * syy - sxy * sxy
    tr, tra = sxx + syy + szz, axx + ayy + azz
    w = np.array([axx * c[0] + axy * c[1] + axz * c[2], axy * c[0] + ayy * c[1] + ayz * c[2], axz * c[0] + ayz * c[1] + azz * c[2]], dtype)
    det = sxx * axx + sxy * axy + sxz * axz
    ww = (w * w).sum(dtype=dtype)
    ok = bool(tra > dtype(1e-2) * (tr * tr) and ww > 0)
    iw = dtype(1) / np.sqrt(ww) if ww > 0 else dtype(0)
    return ok, np.append(-w * iw, (det * dtype(0.2) + (c * w).sum(dtype=dtype)) * iw)


def test_closed_form_plane_is_the_least_squares_solution(oracle):
    """Round 5: the production build forms (pa, pb, pc, pd) of surfOptimization (:783-791) in closed form — n = -5 S^-1 c / (1 + 5 c^T S^-1 c)
    with c the centroid and S the scatter matrix of the five neighbours — instead of through Eigen's column-pivoted QR.  (1) In float64 the
    formula IS the normalised least-squares solution of [p_j] n = -1 (numpy lstsq), for patches anywhere between the origin and 120 m;
    (2) in float32, operation for operation as the kernel does it, the coefficients agree with the oracle's float QR to the QR's own
    conditioning (a few 1e-4 for far patches) and are CLOSER to the float64 solution than the oracle's; (3) neighbourhoods close to one
    line and coincident points are handed to the QR (ok = False)."""
    L = oracle.lib()
    p = oracle.default_params(1)
    rng = np.random.default_rng(11)
    worse = 0; n_cmp = 0
    for trial in range(400):
        centre = rng.normal(0, 1, 3); centre *= rng.uniform(1.0, 120.0) / np.linalg.norm(centre)
        nrm = rng.normal(0, 1, 3); nrm /= np.linalg.norm(nrm)
        if abs(nrm @ centre) < 0.5: continue                    # (a plane through the origin has no solution n . p = -1: not this test's business)
        u = np.cross(nrm, [1.0, 0, 0]); u /= np.linalg.norm(u); v = np.cross(nrm, u)
        spread = rng.uniform(0.05, 0.5)
        nb64 = centre + np.outer(rng.uniform(-spread, spread, 5), u) + np.outer(rng.uniform(-spread, spread, 5), v) + np.outer(rng.normal(0, 0.01, 5), nrm)
        nb = nb64.astype(f32)
        ok64, ref = _plane_closed(nb.astype(np.float64), np.float64)
        sol = np.linalg.lstsq(nb.astype(np.float64), -np.ones(5), rcond=None)[0]
        want = np.append(sol, 1.0) / np.linalg.norm(sol)
        assert np.abs(ref - want).max() <= 1e-7 * max(1.0, np.linalg.norm(centre))            # (1)
        ok32, got = _plane_closed(nb, f32)
        if not ok32: continue
        # (2) the closed form in float: the normal to 2e-4, and the point-to-plane distance of a query ON the patch to 2e-5 m (pd alone is the
        # plane's offset at the ORIGIN, tens of metres away: it moves with the normal's last digits, the distance near the patch does not)
        q64 = nb64.mean(0) + 0.05 * nrm
        assert np.abs(got[:3].astype(np.float64) - want[:3]).max() <= 2e-4, (trial, got, want)
        assert abs((got[:3].astype(np.float64) @ q64 + float(got[3])) - (want[:3] @ q64 + want[3])) <= 2e-5, (trial, got, want)
        # the oracle's float QR on the same neighbours: its coefficients for a query on the centroid's normal
        q = q64.astype(f32)
        cf = np.zeros(4, f32)
        if L.orc_surf_coeff(fp(nb), fp(q), 1.0, C.byref(p), fp(cf)) != 1: continue
        s_o = np.linalg.norm(cf[:3])
        n_o = cf[:3].astype(np.float64) / s_o
        n_cmp += 1
        e_o = np.abs(n_o - want[:3]).max(); e_c = np.abs(got[:3].astype(np.float64) - want[:3]).max()
        assert e_o <= 2e-3 and np.abs(n_o - got[:3]).max() <= 2e-3
        worse += e_c > e_o + 1e-7
    assert n_cmp >= 200 and worse <= n_cmp // 4, (worse, n_cmp)       # the closed form is (almost always) at least as close to the exact solution
    # (3)
    line = (np.array([20.0, 5.0, 1.0]) + np.outer(np.linspace(-0.4, 0.4, 5), [1.0, 0.2, 0.0]) + rng.normal(0, 1e-3, (5, 3))).astype(f32)
    assert not _plane_closed(line, f32)[0]
    assert not _plane_closed(np.tile(np.array([[3.0, 4.0, 5.0]], f32), (5, 1)), f32)[0]


def test_jacobian_row_matches_finite_differences(oracle):
    """Row = d(coeff . (R(T) p + t))/dT in the order [roll, pitch, yaw, x, y, z] (LMOptimization :889-915)."""
    L = oracle.lib()
    rng = np.random.default_rng(2)
    for _ in range(30):
        T = np.concatenate([rng.uniform(-0.5, 0.5, 3), rng.uniform(-5, 5, 3)]).astype(f32)
        ori = rng.uniform(-20, 20, 3).astype(f32)
        n = rng.normal(size=3); n /= np.linalg.norm(n)
        cf = np.array([n[0], n[1], n[2], 0.123], f32)
        row, b = np.zeros(6, f32), np.zeros(1, f32)
        L.orc_jacobian_row(fp(T), fp(ori), fp(cf), fp(row), fp(b))
        assert b[0] == -cf[3]

        def resid(Td):
            M = np.zeros(12, f32)
            L.orc_pose_to_matrix(fp(Td.astype(f32)), fp(M))
            M = M.reshape(3, 4).astype(np.float64)
            return float(cf[:3].astype(np.float64) @ (M[:, :3] @ ori.astype(np.float64) + M[:, 3]))
        num = np.zeros(6)
        for k in range(6):
            h = 1e-3
            Tp, Tm = T.astype(np.float64).copy(), T.astype(np.float64).copy()
            Tp[k] += h; Tm[k] -= h
            num[k] = (resid(Tp) - resid(Tm)) / (2 * h)
        assert np.allclose(row, num, atol=5e-3 * max(1.0, np.abs(num).max()))


def test_transform_update_blend_and_clamp(oracle):
    L = oracle.lib()
    p = oracle.default_params(1)
    imu = oracle.Imu(1, 0.10, -0.06)
    T = np.array([0.02, 0.03, 1.0, 1, 2, 3], f32)
    L.orc_transform_update(C.byref(p), C.byref(imu), fp(T))
    # slerp of single-axis rotations = linear blend of the angle with weight imuRPYWeight = 0.1
    assert np.allclose(T[:2], [0.02 + 0.1 * (0.10 - 0.02), 0.03 + 0.1 * (-0.06 - 0.03)], atol=1e-6)
    assert np.array_equal(T[2:], np.array([1.0, 1, 2, 3], f32))
    p.rotation_tol, p.z_tol = 0.01, 2.5
    T = np.array([0.5, -0.5, 1.0, 1, 2, 3], f32)
    L.orc_transform_update(C.byref(p), None, fp(T))
    assert np.allclose(T, [0.01, -0.01, 1.0, 1, 2, 2.5])
    # |imuPitchInit| >= 1.4 disables the blend (:980); variant #3 never blends
    p = oracle.default_params(1)
    T = np.array([0.02, 0.03, 0, 0, 0, 0], f32)
    L.orc_transform_update(C.byref(p), C.byref(oracle.Imu(1, 0.3, 1.45)), fp(T))
    assert np.allclose(T[:2], [0.02, 0.03])
    p3 = oracle.default_params(3)
    L.orc_transform_update(C.byref(p3), C.byref(imu), fp(T))
    assert np.allclose(T[:2], [0.02, 0.03])
