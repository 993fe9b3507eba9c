"""pcl::IterativeClosestPoint as the loop-closure code drives it (subMapOptmizationNode.cpp:2763-2833).

CPU: the oracle's Umeyama step against numpy's SVD (Kabsch) and its ICP loop against a plain numpy/cKDTree loop.
GPU: liblisreg against the oracle (double-accumulated means: see test_oracle_float_sum_noise for why the reference's float
running sums cannot be a parity target) — a floating-point path: the final transformation within 1e-3 m / 1e-3 rad, the same convergence state and iteration count, the fitness score within 2e-3 relative."""
import numpy as np
import pytest

f32 = np.float32
DBL_MAX = np.finfo(np.float64).max


def _cat(a, b):
    o = np.zeros(len(a) + len(b), a.dtype)
    o[: len(a)], o[len(a):] = a, b
    return o


def _case(seed, n_map=40000, trans=0.6, rot_deg=3.0, hw=(32, 900)):
    """target = submap (map frame); source = a scan of the same scene placed in the map frame with a pose error"""
    from lisreg import synth
    mc, ms = synth.make_submap(n_map, seed=seed, labelled=True)
    tgt = _cat(mc, ms)
    sc = synth.make_scan(hw[0], hw[1], seed + 1, labelled=True)
    src = _cat(sc["corner"], sc["surf"])
    rng = np.random.default_rng(seed)
    T_bad = synth.perturb_pose(sc["T_true"], rng, trans=trans, rot_deg=rot_deg)
    M = synth.pose_matrix(T_bad)
    w = synth.pcl_xyz(src).astype(np.float64) @ M[:3, :3].T + M[:3, 3]
    src["x"], src["y"], src["z"] = w[:, 0].astype(f32), w[:, 1].astype(f32), w[:, 2].astype(f32)
    # the correction ICP should find: moves the mis-placed scan onto the map
    M_fix = synth.pose_matrix(sc["T_true"]) @ np.linalg.inv(M)
    return tgt, src, M_fix


def _rot_angle(R):
    v = 0.5 * np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])      # well conditioned near 0, unlike arccos(trace)
    return float(np.arctan2(np.linalg.norm(v), (np.trace(R) - 1) / 2))


def _pose_diff(A, B):
    A, B = np.asarray(A, np.float64), np.asarray(B, np.float64)
    return _rot_angle(A[:3, :3].T @ B[:3, :3]), float(np.abs(A[:3, 3] - B[:3, 3]).max())


# ---------------------------------------------------------------- CPU
def test_oracle_umeyama_matches_kabsch(oracle):
    rng = np.random.default_rng(1)
    for planar in (False, True):
        a = rng.normal(0, 5, (500, 3))
        if planar:
            a[:, 2] = 0.0                                             # rank-2 covariance
        ang = np.array([0.2, -0.1, 0.7])
        from lisreg import synth
        M = synth.pose_matrix([*ang, 1.0, -2.0, 0.5])
        b = a @ M[:3, :3].T + M[:3, 3] + rng.normal(0, 0.01, a.shape)
        T = oracle.umeyama(a.astype(f32), b.astype(f32))
        ac, bc = a - a.mean(0), b - b.mean(0)
        U, S, Vt = np.linalg.svd(bc.T @ ac)
        D = np.diag([1, 1, np.sign(np.linalg.det(U) * np.linalg.det(Vt))])
        R = U @ D @ Vt
        t = b.mean(0) - R @ a.mean(0)
        assert np.abs(T[:3, :3] - R).max() < 2e-6 and np.abs(T[:3, 3] - t).max() < 2e-5
        assert np.array_equal(T[3], [0, 0, 0, 1])
        assert abs(np.linalg.det(T[:3, :3].astype(np.float64)) - 1) < 1e-6
    # reflection case: the optimal orthogonal map is improper, Umeyama must still return a rotation
    a = rng.normal(0, 1, (50, 3)); b = a * np.array([1, 1, -1.0])
    T = oracle.umeyama(a.astype(f32), b.astype(f32))
    assert abs(np.linalg.det(T[:3, :3].astype(np.float64)) - 1) < 1e-5


def _numpy_icp(tgt, src, max_corr, max_iters, eps_t, eps_mse, prev_mse=DBL_MAX):
    from lisreg import synth
    from scipy.spatial import cKDTree
    t = synth.pcl_xyz(tgt).astype(np.float64); cur = synth.pcl_xyz(src).astype(np.float64)
    tree = cKDTree(t)
    F = np.eye(4); iters = 0; state = 0
    while True:
        d, idx = tree.query(cur, k=1)
        ok = d * d <= max_corr * max_corr
        if ok.sum() < 3:
            return F, iters, 5, prev_mse
        a, b = cur[ok], t[idx[ok]]
        ac, bc = a - a.mean(0), b - b.mean(0)
        U, S, Vt = np.linalg.svd(bc.T @ ac)
        R = U @ np.diag([1, 1, np.sign(np.linalg.det(U) * np.linalg.det(Vt))]) @ Vt
        tr = b.mean(0) - R @ a.mean(0)
        cur = cur @ R.T + tr
        Tm = np.eye(4); Tm[:3, :3] = R; Tm[:3, 3] = tr
        F = Tm @ F; iters += 1
        if iters >= max_iters:
            return F, iters, 1, prev_mse
        if 0.5 * (np.trace(R) - 1) >= 1 - eps_t and tr @ tr <= eps_t:
            return F, iters, 2, prev_mse
        mse = float((d[ok] ** 2).mean())
        if abs(mse - prev_mse) < 1e-12:
            return F, iters, 3, prev_mse
        if abs(mse - prev_mse) / prev_mse < eps_mse:
            return F, iters, 4, prev_mse
        prev_mse = mse


@pytest.mark.parametrize("kind,seed,trans,rot", [(0, 41, 0.6, 3.0), (0, 42, 1.5, 6.0), (1, 43, 0.05, 0.3)])
def test_oracle_icp_matches_numpy_loop(oracle, kind, seed, trans, rot):
    tgt, src, M_fix = _case(seed, n_map=15000, trans=trans, rot_deg=rot, hw=(16, 450))
    p = oracle.icp_default_params(kind)
    r = oracle.icp_align(tgt, src, p, float_sums=True)
    F, iters, state, prev = _numpy_icp(tgt, src, p.max_corr_dist, p.max_iters, p.transformation_epsilon, p.euclidean_fitness_epsilon)
    assert r["converged"] and r["state"] == state and r["iters"] == iters
    dr, dt = _pose_diff(r["T"], F)
    assert dr < 2e-4 and dt < 2e-3
    if kind == 0:                                                     # wide gate: recovers the pose error
        dr, dt = _pose_diff(r["T"], M_fix)
        assert dr < 0.01 and dt < 0.1
    assert 0 < r["fitness"] < 1.0


def test_oracle_float_sum_noise(oracle):
    """Eigen's umeyama sums the means sequentially in float; at scan size that is ~1e-4..1e-3 m of order-dependent noise per
    estimate.  The GPU's parity target is the double-mean variant (which tracks an all-double numpy loop to ~1e-5 m); this test
    pins how far the float-sum restatement of the reference sits from it."""
    tgt, src, _ = _case(45, n_map=60000, trans=1.0, rot_deg=4.0, hw=(32, 900))
    p = oracle.icp_default_params(0)
    rf, rd = oracle.icp_align(tgt, src, p, float_sums=True), oracle.icp_align(tgt, src, p, float_sums=False)
    F, iters, state, _ = _numpy_icp(tgt, src, p.max_corr_dist, p.max_iters, p.transformation_epsilon, p.euclidean_fitness_epsilon)
    assert rd["iters"] == iters and rd["state"] == state
    dr, dt = _pose_diff(rd["T"], F)
    assert dr < 5e-5 and dt < 2e-4
    dr, dt = _pose_diff(rf["T"], rd["T"])
    assert dr < 1e-3 and dt < 1e-2                                    # the reference's own summation noise stays below 1 cm
    assert abs(rf["iters"] - rd["iters"]) <= 1


def test_oracle_icp_edges(oracle):
    tgt, src, _ = _case(44, n_map=5000, hw=(8, 240))
    p = oracle.icp_default_params(0)
    far = src.copy(); far["x"] += 1000.0
    r = oracle.icp_align(tgt, far, p)
    assert not r["converged"] and r["state"] == 5 and r["iters"] == 0 and np.array_equal(r["T"], np.eye(4, dtype=f32))
    r = oracle.icp_align(tgt, src[:2], p)                              # < 3 correspondences
    assert not r["converged"] and r["state"] == 5
    p.max_iters = 1
    r = oracle.icp_align(tgt, src, p)
    assert r["converged"] and r["state"] == 1 and r["iters"] == 1
    # the carried correspondences_prev_mse_: a second align() on a `static` object can stop on the relative-MSE test at once
    p = oracle.icp_default_params(0)
    r1 = oracle.icp_align(tgt, src, p)
    p.prev_mse = r1["prev_mse"]
    assert r1["prev_mse"] < DBL_MAX
    g = np.eye(4, dtype=f32)
    r2 = oracle.icp_align(tgt, src, p, guess=g)
    assert r2["converged"]


# ---------------------------------------------------------------- GPU
def _check(rg, ro):
    assert rg["converged"] == ro["converged"] and rg["state"] == ro["state"], (rg, ro)
    assert abs(rg["iters"] - ro["iters"]) <= 1, (rg["iters"], ro["iters"])
    dr, dt = _pose_diff(rg["T"], ro["T"])
    assert dr < 1e-3 and dt < 1e-3, (dr, dt)      # the bar; measured ~1e-6 rad / ~1e-5 m converged, 2e-4 m when cut off mid-flight
    assert abs(rg["fitness"] - ro["fitness"]) <= 2e-3 * ro["fitness"] + 1e-7         # d(fitness) ~ 2 * rms distance * d(pose)
    assert abs(rg["n_corr_last"] - ro["n_corr_last"]) <= max(3, ro["n_corr_last"] // 2000)


@pytest.mark.gpu
@pytest.mark.parametrize("kind,seed,trans,rot,n_map", [(0, 51, 0.6, 3.0, 40000), (0, 52, 1.5, 6.0, 120000), (1, 53, 0.05, 0.3, 40000),
                                                      (0, 54, 3.0, 10.0, 40000)])
def test_hip_icp_matches_oracle(oracle, gpu_ctx, kind, seed, trans, rot, n_map):
    import lisreg
    tgt, src, M_fix = _case(seed, n_map=n_map, trans=trans, rot_deg=rot)
    gpu_ctx.map_index_set(9, tgt)
    po, pg = oracle.icp_default_params(kind), lisreg.icp_default_params(kind)
    assert all(getattr(po, f) == getattr(pg, f) for f, _ in pg._fields_)
    ro = oracle.icp_align(tgt, src, po)
    rg = gpu_ctx.icp_align(9, src, pg, want_aligned=True)
    _check(rg, ro)
    # `output` of align(): the source under the final transformation, other fields copied
    from lisreg import synth
    w = synth.pcl_xyz(src).astype(np.float64) @ rg["T"][:3, :3].astype(np.float64).T + rg["T"][:3, 3]
    assert np.abs(synth.pcl_xyz(rg["aligned"]) - w).max() < 1e-4
    assert np.array_equal(rg["aligned"]["label"], src["label"]) and np.array_equal(rg["aligned"]["intensity"], src["intensity"])


@pytest.mark.gpu
def test_hip_icp_edges_guess_and_device(oracle, gpu_ctx):
    import lisreg
    tgt, src, _ = _case(55, n_map=30000, hw=(16, 450))
    gpu_ctx.map_index_set(9, tgt)
    po, pg = oracle.icp_default_params(0), lisreg.icp_default_params(0)
    far = src.copy(); far["x"] += 1000.0
    rg = gpu_ctx.icp_align(9, far, pg)
    assert not rg["converged"] and rg["state"] == lisreg.ICP_NO_CORRESPONDENCES and rg["iters"] == 0
    assert np.array_equal(rg["T"], np.eye(4, dtype=f32))
    _check(gpu_ctx.icp_align(9, src[:2], pg), oracle.icp_align(tgt, src[:2], po))
    rg = gpu_ctx.icp_align(9, src[:0], pg)
    assert not rg["converged"] and rg["fitness"] == DBL_MAX
    po.max_iters = pg.max_iters = 1
    _check(gpu_ctx.icp_align(9, src, pg), oracle.icp_align(tgt, src, po))
    po.max_iters = pg.max_iters = 7
    _check(gpu_ctx.icp_align(9, src, pg), oracle.icp_align(tgt, src, po))
    # initial guess + carried prev_mse
    from lisreg import synth
    g = synth.pose_matrix([0.01, -0.02, 0.03, 0.2, -0.1, 0.05]).astype(f32)
    po, pg = oracle.icp_default_params(0), lisreg.icp_default_params(0)
    ro, rg = oracle.icp_align(tgt, src, po, guess=g), gpu_ctx.icp_align(9, src, pg, guess=g)
    _check(rg, ro)
    po.prev_mse, pg.prev_mse = ro["prev_mse"], ro["prev_mse"]
    _check(gpu_ctx.icp_align(9, src, pg), oracle.icp_align(tgt, src, po))
    with pytest.raises(lisreg.LisregError):
        gpu_ctx.icp_align(77, src, pg)
    # device records
    rs = lisreg.pack_device_records(src)
    ds, dout = lisreg.DeviceArray(rs), lisreg.DeviceArray(np.zeros_like(rs))
    pg = lisreg.icp_default_params(0)
    rd = gpu_ctx.icp_align_device(9, ds.ptr, len(rs), pg, out_ptr=dout.ptr)
    rh = gpu_ctx.icp_align(9, src, pg, want_aligned=True)
    assert np.array_equal(rd["T"], rh["T"]) and rd["iters"] == rh["iters"] and rd["fitness"] == rh["fitness"]
    got = lisreg.device_to_host(dout.ptr, rs.shape, np.float32)
    assert np.array_equal(got[:, :3], synth.pcl_xyz(rh["aligned"])) and np.array_equal(got[:, 3].view(np.uint32), rs[:, 3].view(np.uint32))


# ---------------------------------------------------------------- the candidate loop as one call (lisreg_icp_align_batch)
def _same(a, b):
    return (np.array_equal(a["T"], b["T"]) and a["iters"] == b["iters"] and a["state"] == b["state"] and a["converged"] == b["converged"]
            and a["n_corr_last"] == b["n_corr_last"] and a["fitness"] == b["fitness"] and a["prev_mse"] == b["prev_mse"])


@pytest.mark.gpu
def test_hip_icp_batch_equals_single_calls_bitwise_and_oracle(oracle, gpu_ctx):
    """Loop-closure verification (subMapOptmizationNode.cpp:2776-2840): candidates with their OWN targets, sources and guesses in one
    call — each result equal to lisreg_icp_align on that candidate alone TO THE BIT (one lane per query in the batch, four alone: the
    sums are formed by the same tree), and within the ICP bar of the oracle."""
    import lisreg
    from lisreg import synth
    cases = [_case(61, n_map=30000, trans=0.5, rot_deg=2.0, hw=(16, 450)), _case(62, n_map=50000, trans=1.0, rot_deg=4.0, hw=(32, 900)),
             _case(63, n_map=20000, trans=0.2, rot_deg=1.0, hw=(16, 450))]
    for k, (tgt, _, _) in enumerate(cases):
        gpu_ctx.map_index_set(20 + k, tgt)
    g1 = synth.pose_matrix([0.01, -0.02, 0.03, 0.2, -0.1, 0.05]).astype(f32)
    far = cases[0][1].copy(); far["x"] += 1000.0
    items = [(20, cases[0][1], None), (21, cases[1][1], g1), (22, cases[2][1], None), (21, cases[0][1][:2], None), (20, far, None),
             (22, cases[2][1][:0], None), (20, cases[0][1], None)]
    pg, po = lisreg.icp_default_params(0), oracle.icp_default_params(0)
    rb = gpu_ctx.icp_align_batch(items, pg)
    assert len(rb) == len(items)
    for (slot, src, g), r in zip(items, rb):
        rs = gpu_ctx.icp_align(slot, src, pg, guess=g)
        assert _same(r, rs), (slot, len(src), r, rs)
        if len(src) > 2 and r["state"] != lisreg.ICP_NO_CORRESPONDENCES:
            _check(r, oracle.icp_align(cases[slot - 20][0], src, po, guess=g))
    assert _same(rb[0], rb[6])                                             # the same candidate twice
    assert rb[4]["state"] == lisreg.ICP_NO_CORRESPONDENCES and rb[5]["fitness"] == DBL_MAX and not rb[5]["converged"]
    # a bigger batch switches to one lane per query (>= 500 k source points): still the same bits
    many = [(20 + (k % 3), cases[k % 3][1], None) for k in range(45)]
    rm = gpu_ctx.icp_align_batch(many, pg)
    assert sum(len(s) for _, s, _ in many) >= 500000
    for k in range(45):
        assert _same(rm[k], rm[k % 3]), k
    assert _same(rm[0], rb[0]) and _same(rm[2], rb[2])
    # device records
    recs = [lisreg.pack_device_records(cases[k][1]) for k in range(3)]
    devs = [lisreg.DeviceArray(r) for r in recs]
    rd = gpu_ctx.icp_align_batch([(20 + k, (devs[k].ptr, len(recs[k])), None) for k in range(3)], pg)
    assert _same(rd[0], rb[0]) and _same(rd[2], rb[2])
    with pytest.raises(lisreg.LisregError):
        gpu_ctx.icp_align_batch([(77, cases[0][1], None)], pg)
    assert gpu_ctx.icp_align_batch([], pg) == []


@pytest.mark.gpu
def test_hip_icp_flat_search_resolves_exact_ties_like_the_column_walk(gpu_ctx, monkeypatch):
    """The one-lane-per-query ICP kernels search with the flattened walk (nn1_search_flat: strict insertion of a group's unique minimum,
    the full (distance, original index) rule only on equality), the four-lane ones with the column walk.  A target on a lattice and a
    source on its cell centres make EVERY query's nearest neighbour an eight-way exact tie between different points, so the rule decides
    every correspondence: both forms must give the same alignment to the bit (and the same again with the lattice's points listed twice)."""
    import lisreg
    g = np.arange(-6, 7, dtype=f32) * f32(0.25)
    X, Y, Z = np.meshgrid(g, g, g[:5], indexing="ij")
    lat = np.stack([X.ravel(), Y.ravel(), Z.ravel()], 1).astype(f32)
    rng = np.random.default_rng(5)
    lat = lat[rng.permutation(len(lat))]                                    # original indices in no spatial order
    cases_dtype = _case(61, n_map=2000, hw=(16, 450))[0].dtype
    def cloud(xyz):
        c = np.zeros(len(xyz), dtype=cases_dtype)
        c["x"], c["y"], c["z"] = xyz[:, 0], xyz[:, 1], xyz[:, 2]
        return c
    tgt = cloud(lat)
    tgt2 = cloud(np.concatenate([lat, lat[::-1]]))
    centres = (lat[np.all(lat < 1.4, axis=1) & (lat[:, 2] < 0.9)] + f32(0.125)).astype(f32)
    src = cloud(centres[rng.permutation(len(centres))])
    gpu_ctx.map_index_set(40, tgt); gpu_ctx.map_index_set(41, tgt2)
    pg = lisreg.icp_default_params(0)
    items = [(40, src, None), (41, src, None), (40, src[: len(src) // 2], None)]
    out = {}
    for q in ("1", "4"):
        monkeypatch.setenv("LISREG_NN1_Q", q)
        out[q] = gpu_ctx.icp_align_batch(items, pg)
    monkeypatch.delenv("LISREG_NN1_Q")
    for a, b in zip(out["1"], out["4"]):
        assert _same(a, b), (a, b)
        assert a["n_corr_last"] > 0


@pytest.mark.gpu
def test_map_index_set_batch_equals_single_sets(gpu_ctx):
    """setInputTarget of all candidates in one call (lisreg_map_index_set_batch): k = 1 queries and ICP alignments against the maps it
    builds equal those against maps built one by one — host clouds, device records, an empty and a one-point cloud among them."""
    import lisreg
    from lisreg import synth
    cases = [_case(71 + k, n_map=int(15000 + 9000 * k), trans=0.3, rot_deg=1.5, hw=(16, 450)) for k in range(5)]
    holes = cases[3][0].copy()                                                  # a cloud with non-finite points in it (never anyone's neighbour)
    holes["x"][5::97] = np.nan; holes["z"][11::131] = np.nan
    clouds = [c[0] for c in cases] + [cases[0][0][:0], cases[1][0][:1], holes]
    for k, cl in enumerate(clouds):
        gpu_ctx.map_index_set(60 + k, cl)
    gpu_ctx.map_index_set_batch([80 + k for k in range(len(clouds))], clouds)
    pg = lisreg.icp_default_params(0)
    q = cases[2][1][:4000]
    for k in range(len(clouds)):
        i1, d1 = gpu_ctx.nearest(60 + k, q, 5.0)
        i2, d2 = gpu_ctx.nearest(80 + k, q, 5.0)
        assert np.array_equal(i1, i2) and np.array_equal(d1.view(np.uint32), d2.view(np.uint32)), k
    for k in range(5):
        assert _same(gpu_ctx.icp_align(60 + k, cases[k][1], pg), gpu_ctx.icp_align(80 + k, cases[k][1], pg)), k
    # device records, re-using slots (a second batch over the same slots replaces the maps)
    recs = [lisreg.DeviceArray(lisreg.pack_device_records(c[0])) for c in cases]
    gpu_ctx.map_index_set_batch([80 + k for k in range(5)], [(r.ptr, r.shape[0]) for r in reversed(recs)])
    for k in range(5):
        i1, d1 = gpu_ctx.nearest(60 + (4 - k), q, 5.0)
        i2, d2 = gpu_ctx.nearest(80 + k, q, 5.0)
        assert np.array_equal(i1, i2) and np.array_equal(d1.view(np.uint32), d2.view(np.uint32)), k
    with pytest.raises(lisreg.LisregError):
        gpu_ctx.map_index_set_batch([90, 90], clouds[:2])
    # the batch takes the strip form of the index build where the grids fit it (round 5); index_build 0 keeps the batched bucket sort: same maps
    gpu_ctx.set_option("index_build", 0)
    try:
        gpu_ctx.map_index_set_batch([100 + k for k in range(len(clouds))], clouds)
    finally:
        gpu_ctx.set_option("index_build", 2)
    for k in range(len(clouds)):
        i1, d1 = gpu_ctx.nearest(60 + k, q, 5.0)
        i2, d2 = gpu_ctx.nearest(100 + k, q, 5.0)
        assert np.array_equal(i1, i2) and np.array_equal(d1.view(np.uint32), d2.view(np.uint32)), k


@pytest.mark.gpu
def test_hip_icp_batch_chained_prev_mse_follows_the_static_object(oracle, gpu_ctx):
    """The reference's ICP object is `static` (subMapOptmizationNode.cpp:2763): correspondences_prev_mse_ of candidate k - 1 is what
    candidate k's first MSE comparison sees.  chain_prev_mse = 1 must give what a sequential loop gives — including the case where that
    first comparison stops the alignment (a candidate that starts where its predecessor ended)."""
    import lisreg
    tgt, src, _ = _case(64, n_map=30000, trans=0.4, rot_deg=2.0, hw=(16, 450))
    tgt2, src2, _ = _case(65, n_map=30000, trans=0.3, rot_deg=1.0, hw=(16, 450))
    gpu_ctx.map_index_set(30, tgt); gpu_ctx.map_index_set(31, tgt2)
    pg = lisreg.icp_default_params(0)
    r0 = gpu_ctx.icp_align(30, src, pg)
    assert r0["converged"]
    items = [(30, src, None), (30, src, r0["T"]), (31, src2, None), (30, src[:2], None), (31, src2, None)]
    # sequential loop through the single entry point, feeding prev_mse forward
    seq, p = [], lisreg.icp_default_params(0)
    for slot, s, g in items:
        r = gpu_ctx.icp_align(slot, s, p, guess=g)
        seq.append(r)
        p.prev_mse = r["prev_mse"]
    rb = gpu_ctx.icp_align_batch(items, pg, chain_prev_mse=True)
    for k in range(len(items)):
        assert _same(rb[k], seq[k]), (k, rb[k], seq[k])
    # the second candidate starts at the first one's answer: its first MSE is within the relative bound of the carried one -> one iteration
    assert seq[1]["iters"] == 1 and seq[1]["state"] in (lisreg.ICP_REL_MSE, lisreg.ICP_ABS_MSE, lisreg.ICP_TRANSFORM)
    ind = gpu_ctx.icp_align_batch(items, pg, chain_prev_mse=False)
    assert _same(ind[0], seq[0]) and _same(ind[2], gpu_ctx.icp_align(31, src2, pg))
    # and the oracle driven the same way
    po = oracle.icp_default_params(0)
    tg = {30: tgt, 31: tgt2}
    for k, (slot, s, g) in enumerate(items):
        ro = oracle.icp_align(tg[slot], s, po, guess=g)
        if len(s) > 2:
            _check(rb[k], ro)
        po.prev_mse = ro["prev_mse"]


# ---------------------------------------------------------------- OptimizedICPGN (registration.cpp:19-115)
def _numpy_icp_gn(tgt, src, iters, max_corr, T0):
    """all-double Gauss-Newton point-to-point ICP with the reference's update rule (t += d[:3], R = R exp(d[3:]))"""
    from lisreg import synth
    from scipy.spatial import cKDTree
    from scipy.spatial.transform import Rotation
    t = synth.pcl_xyz(tgt).astype(np.float64); p = synth.pcl_xyz(src).astype(np.float64)
    tree = cKDTree(t)
    T = np.array(T0, np.float64)
    for _ in range(iters):
        tp = p @ T[:3, :3].T + T[:3, 3]
        d, idx = tree.query(tp, k=1)
        ok = d * d <= max_corr                                     # squared distance vs un-squared threshold, as the reference
        e = (tp - t[idx])[ok]; po = p[ok]
        hat = np.zeros((len(po), 3, 3))
        hat[:, 0, 1], hat[:, 0, 2], hat[:, 1, 0], hat[:, 1, 2], hat[:, 2, 0], hat[:, 2, 1] = -po[:, 2], po[:, 1], po[:, 2], -po[:, 0], -po[:, 1], po[:, 0]
        A = -np.einsum("ij,njk->nik", T[:3, :3], hat)
        J = np.concatenate([np.broadcast_to(np.eye(3), A.shape), A], 2)
        H = np.einsum("nra,nrb->ab", J, J); B = -np.einsum("nra,nr->a", J, e)
        if np.linalg.det(H) == 0:
            continue
        dx = np.linalg.solve(H, B)
        T[:3, 3] += dx[:3]
        T[:3, :3] = T[:3, :3] @ Rotation.from_rotvec(dx[3:]).as_matrix()
    tp = p @ T[:3, :3].T + T[:3, 3]
    d, _ = tree.query(tp, k=1)
    return T, float((d * d).mean())


@pytest.mark.parametrize("seed,trans,rot,iters,max_corr", [(61, 0.5, 2.0, 12, 4.0), (62, 1.0, 4.0, 20, 25.0)])
def test_oracle_icp_gn_matches_numpy(oracle, seed, trans, rot, iters, max_corr):
    tgt, src, M_fix = _case(seed, n_map=15000, trans=trans, rot_deg=rot, hw=(16, 450))
    T0 = np.eye(4, dtype=f32)
    r = oracle.icp_gn_match(tgt, src, iters, max_corr, T0)
    T, fit = _numpy_icp_gn(tgt, src, iters, max_corr, T0)
    assert r["steps_applied"] == iters
    dr, dt = _pose_diff(r["T"], T)
    assert dr < 2e-4 and dt < 2e-3, (dr, dt)          # float 6x6 inverse of a cond ~1e5 Hessian (as the reference) vs an all-double loop
    assert abs(r["fitness"] - fit) < 5e-3 * fit
    dr, dt = _pose_diff(r["T"], M_fix)
    assert dr < 0.01 and dt < 0.2                     # sanity only: point-to-point on differently sampled clouds has a biased optimum
    rf = oracle.icp_gn_match(tgt, src, iters, max_corr, T0, float_sums=True)          # the reference's float running sums
    dr, dt = _pose_diff(rf["T"], r["T"])
    assert dr < 1e-3 and dt < 1e-2


def test_oracle_icp_gn_edges(oracle):
    tgt, src, _ = _case(63, n_map=5000, hw=(8, 240))
    T0 = np.eye(4, dtype=f32)
    r = oracle.icp_gn_match(tgt, src, 0, 4.0, T0)
    assert r["steps_applied"] == 0 and np.array_equal(r["T"], T0) and r["fitness"] > 0
    far = src.copy(); far["x"] += 1000.0
    r = oracle.icp_gn_match(tgt, far, 5, 4.0, T0)                                      # no correspondences: det(H) == 0, T untouched
    assert r["steps_applied"] == 0 and r["n_corr_last"] == 0 and np.array_equal(r["T"], T0)
    bad = src.copy(); bad["x"][::7] = np.nan                                             # pcl::isFinite filter
    r1, r2 = oracle.icp_gn_match(tgt, bad, 3, 4.0, T0), oracle.icp_gn_match(tgt, np.delete(src, np.s_[::7]), 3, 4.0, T0)
    assert np.array_equal(r1["T"], r2["T"]) and r1["n_corr_last"] == r2["n_corr_last"]


@pytest.mark.gpu
@pytest.mark.parametrize("seed,trans,rot,iters,max_corr,n_map", [(71, 0.5, 2.0, 12, 4.0, 40000), (72, 1.0, 4.0, 20, 25.0, 120000)])
def test_hip_icp_gn_matches_oracle(oracle, gpu_ctx, seed, trans, rot, iters, max_corr, n_map):
    tgt, src, _ = _case(seed, n_map=n_map, trans=trans, rot_deg=rot)
    gpu_ctx.map_index_set(10, tgt)
    from lisreg import synth
    T0 = synth.pose_matrix([0.0, 0.0, 0.01, 0.05, -0.05, 0.0]).astype(f32)
    ro = oracle.icp_gn_match(tgt, src, iters, max_corr, T0)
    rg = gpu_ctx.icp_gn_match(10, src, iters, max_corr, T0, want_transformed=True)
    assert rg["steps_applied"] == ro["steps_applied"] == iters
    dr, dt = _pose_diff(rg["T"], ro["T"])
    assert dr < 1e-3 and dt < 1e-3, (dr, dt)
    assert abs(rg["fitness"] - ro["fitness"]) <= 2e-3 * ro["fitness"]
    assert abs(rg["n_corr_last"] - ro["n_corr_last"]) <= max(3, ro["n_corr_last"] // 2000)
    w = synth.pcl_xyz(src).astype(np.float64) @ rg["T"][:3, :3].astype(np.float64).T + rg["T"][:3, 3]
    assert np.abs(synth.pcl_xyz(rg["transformed"]) - w).max() < 1e-4 and np.array_equal(rg["transformed"]["label"], src["label"])


@pytest.mark.gpu
def test_hip_icp_gn_edges(oracle, gpu_ctx):
    import lisreg
    tgt, src, _ = _case(73, n_map=30000, hw=(16, 450))
    gpu_ctx.map_index_set(10, tgt)
    T0 = np.eye(4, dtype=f32)
    r = gpu_ctx.icp_gn_match(10, src, 0, 4.0, T0)
    assert r["steps_applied"] == 0 and np.array_equal(r["T"], T0)
    assert abs(r["fitness"] - oracle.icp_gn_match(tgt, src, 0, 4.0, T0)["fitness"]) < 1e-5
    far = src.copy(); far["x"] += 1000.0
    r = gpu_ctx.icp_gn_match(10, far, 5, 4.0, T0)
    assert r["steps_applied"] == 0 and r["n_corr_last"] == 0 and np.array_equal(r["T"], T0)
    bad = src.copy(); bad["x"][::7] = np.nan
    r1, r2 = gpu_ctx.icp_gn_match(10, bad, 3, 4.0, T0), gpu_ctx.icp_gn_match(10, np.delete(src, np.s_[::7]), 3, 4.0, T0)
    assert r1["n_corr_last"] == r2["n_corr_last"] and _pose_diff(r1["T"], r2["T"])[1] < 1e-5
    with pytest.raises(lisreg.LisregError):
        gpu_ctx.icp_gn_match(88, src, 3, 4.0, T0)
