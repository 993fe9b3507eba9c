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
