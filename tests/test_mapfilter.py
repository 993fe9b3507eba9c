"""local-map maintenance — k = 1 search, dynamic-point filter, box crop, cloud bounds.

CPU: the oracle's restatements against independent numpy / scipy computations.
GPU: liblisreg against the oracle — survivors identical (same points, same order, every field), k = 1 squared distances bit
for bit (both accumulate (dx^2 + dy^2) + dz^2 in float without contraction)."""
import numpy as np
import pytest

f32 = np.float32
FLT_MAX = 3.4028234663852886e38


def same_points(got, want):
    return len(got) == len(want) and all(np.array_equal(got[f], want[f]) for f in want.dtype.names)


XYZI_DTYPE = np.dtype({"names": ["x", "y", "z", "intensity"], "formats": ["<f4"] * 4, "offsets": [0, 4, 8, 16], "itemsize": 32})


def _scene(seed, n_map=30000, shift=0.35, labelled=True):
    """A map cloud and a scan of the same scene (map frame) jittered by `shift` m, so nearest distances straddle the thresholds."""
    from lisreg import synth
    mc, ms = synth.make_submap(n_map, seed=seed, labelled=labelled)
    def cat(a, b):                                                    # np.concatenate would re-pack the 32-byte structs
        o = np.zeros(len(a) + len(b), a.dtype)
        o[: len(a)], o[len(a):] = a, b
        return o
    m = cat(mc, ms)
    sc = synth.make_scan(32, 900, seed + 1, labelled=labelled)
    q = cat(sc["corner"], sc["surf"])
    assert m.dtype.itemsize == 32 and q.dtype.itemsize == 32
    M = synth.pose_matrix(sc["T_true"])
    w = synth.pcl_xyz(q).astype(np.float64) @ M[:3, :3].T + M[:3, 3]
    rng = np.random.default_rng(seed)
    w[:, :2] += rng.normal(0, shift, (len(q), 2))
    far = rng.random(len(q)) < 0.02                                   # a few points far from everything
    w[far, 2] += rng.uniform(5, 40, int(far.sum()))
    q["x"], q["y"], q["z"] = w[:, 0].astype(f32), w[:, 1].astype(f32), w[:, 2].astype(f32)
    q["intensity"] = rng.uniform(0, 255, len(q)).astype(f32)
    if not labelled:                                                  # PointXYZI structs
        def strip(c):
            o = np.zeros(len(c), XYZI_DTYPE)
            for f in XYZI_DTYPE.names:
                o[f] = c[f]
            return o
        m, q = strip(m), strip(q)
    return m, q


def _d2_numpy(m, q):
    from lisreg import synth
    from scipy.spatial import cKDTree
    mx, qx = synth.pcl_xyz(m), synth.pcl_xyz(q)
    _, idx = cKDTree(mx.astype(np.float64)).query(qx.astype(np.float64), k=1)
    d = qx - mx[idx]
    return idx, (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]      # float32, L2_Simple order


# ---------------------------------------------------------------- CPU: oracle vs numpy / scipy
def test_oracle_nearest_matches_ckdtree(oracle):
    m, q = _scene(3, n_map=8000)
    idx, d2 = oracle.nearest(m, q)
    ref_idx, ref_d2 = _d2_numpy(m, q)
    # the double-precision tree may pick another of two float-equidistant points; the float distance must agree to an ulp
    assert np.allclose(d2, ref_d2, rtol=3e-7, atol=0)
    assert (idx == ref_idx).mean() > 0.999
    idx_c, d2_c = oracle.nearest(m, q, max_dist=1.0)
    inside = d2 <= 1.0
    assert np.array_equal(idx_c[inside], idx[inside]) and np.all(idx_c[~inside] == -1)


@pytest.mark.parametrize("dmin,dmax,near", [(0.3, 1.0, 0.05), (FLT_MAX, FLT_MAX, 0.1), (0.5, FLT_MAX, 0.0), (0.15, 0.25, 0.15)])
def test_oracle_dynamic_filter_matches_numpy(oracle, dmin, dmax, near):
    m, q = _scene(4, n_map=8000)
    radius = 25.0
    out, applied = oracle.dynamic_filter(m, q, radius, dmin, dmax, near)
    _, d2 = _d2_numpy(m, q)
    r2 = q["x"] * q["x"] + q["y"] * q["y"]
    with np.errstate(over="ignore"):
        n2, mn2, mx2 = f32(near) * f32(near), f32(dmin) * f32(dmin), f32(dmax) * f32(dmax)
    keep = (r2 > f32(radius) * f32(radius)) | ((d2 > n2) & (d2 < mn2)) | (d2 > mx2)
    assert applied and 0 < keep.sum() < len(q)
    assert same_points(out, q[keep])


def test_oracle_dynamic_filter_small_and_empty(oracle):
    m, q = _scene(5, n_map=4000)
    out, applied = oracle.dynamic_filter(m, q[:10], 25.0, 0.3, 1.0, 0.05)          # subMap.h:1071: <= 10 points -> untouched
    assert not applied and same_points(out, q[:10])
    out, applied = oracle.dynamic_filter(m[:0], q, 25.0, 0.3, 1.0, 0.05)
    assert same_points(out, q)


def test_oracle_bbx_and_bounds(oracle):
    from lisreg import synth
    m, q = _scene(6, n_map=4000)
    b = oracle.cloud_bounds(q)
    xyz = synth.pcl_xyz(q).astype(np.float64)
    assert np.array_equal(b, np.concatenate([xyz.min(0), xyz.max(0)]))
    e = oracle.cloud_bounds(q[:0])
    assert np.all(e[:3] == np.finfo(np.float64).max) and np.all(e[3:] == -np.finfo(np.float64).max)
    x17 = float(q["x"][17])                                           # a point ON the face: strict inequality drops it
    box = np.array([x17, -30.0, -2.0, x17 + 25.0, 30.0, 10.0])
    inside = np.all((xyz > box[:3]) & (xyz < box[3:]), axis=1)
    assert 0 < inside.sum() < len(q) and not inside[17]
    assert same_points(oracle.bbx_filter(q, box), q[inside])
    assert same_points(oracle.bbx_filter(q, box, delete_box=True), q[~inside])


# ---------------------------------------------------------------- GPU: liblisreg vs oracle
@pytest.mark.gpu
@pytest.mark.parametrize("seed,n_map", [(21, 30000), (22, 150000)])
def test_hip_nearest_matches_oracle(oracle, gpu_ctx, seed, n_map):
    m, q = _scene(seed, n_map=n_map)
    gpu_ctx.map_index_set(5, m)
    idx, d2 = gpu_ctx.nearest(5, q)
    o_idx, o_d2 = oracle.nearest(m, q)
    assert np.array_equal(d2, o_d2)                                    # bit for bit
    from lisreg import synth
    mx, qx = synth.pcl_xyz(m), synth.pcl_xyz(q)
    d = qx - mx[idx]
    assert np.array_equal((d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2], d2)     # idx really is that point
    assert (idx == o_idx).mean() > 0.999                                # ties may differ (smallest index here)
    for cap in (0.2, 1.0, 3.0):
        idx_c, d2_c = gpu_ctx.nearest(5, q, cap)
        inside = o_d2 <= f32(cap) * f32(cap)
        assert np.array_equal(idx_c >= 0, inside)
        assert np.array_equal(d2_c[inside], o_d2[inside])


@pytest.mark.gpu
def test_hip_nearest_ties_and_outside_grid(oracle, gpu_ctx):
    from lisreg import synth
    xyz = np.array([[0, 0, 0], [2, 0, 0], [0, 2, 0], [2, 2, 0], [1, 1, 5], [2, 0, 0]], f32)     # 1 and 5 coincide
    m = synth.to_pcl(xyz)
    gpu_ctx.map_index_set(6, m)
    q = synth.to_pcl(np.array([[1, 1, 0], [2, 0, 0.1], [500, -300, 40], [1, 1, 2.5], [-80, 0, 0]], f32))
    idx, d2 = gpu_ctx.nearest(6, q)
    assert idx.tolist() == [0, 1, 1, 4, 0]                              # equidistant -> smallest index
    _, o_d2 = oracle.nearest(m, q)
    assert np.array_equal(d2, o_d2)
    idx, _ = gpu_ctx.nearest(6, q, 3.0)
    assert idx.tolist() == [0, 1, -1, 4, -1]
    gpu_ctx.map_index_set(6, m[:0])                                     # empty map: nothing found
    idx, _ = gpu_ctx.nearest(6, q)
    assert np.all(idx == -1)


@pytest.mark.gpu
@pytest.mark.parametrize("dmin,dmax,near", [(0.3, 1.0, 0.05), (FLT_MAX, FLT_MAX, 0.1), (0.5, FLT_MAX, 0.0), (0.15, 0.25, 0.15),
                                            (3.0, 3.1, 0.02)])
@pytest.mark.parametrize("labelled", [True, False])
def test_hip_dynamic_filter_matches_oracle(oracle, gpu_ctx, dmin, dmax, near, labelled):
    m, q = _scene(31, n_map=60000, labelled=labelled)
    gpu_ctx.map_index_set(7, m)
    want, _ = oracle.dynamic_filter(m, q, 30.0, dmin, dmax, near)
    got, applied = gpu_ctx.dynamic_filter(7, q, 30.0, dmin, dmax, near)
    assert applied and 0 < len(want) < len(q)
    assert same_points(got, want)


@pytest.mark.gpu
def test_hip_dynamic_filter_edges_and_device(oracle, gpu_ctx):
    import lisreg
    m, q = _scene(32, n_map=20000)
    gpu_ctx.map_index_set(7, m)
    got, applied = gpu_ctx.dynamic_filter(7, q[:10], 30.0, 0.3, 1.0, 0.05)
    assert not applied and same_points(got, q[:10])
    got, applied = gpu_ctx.dynamic_filter(7, q[:0], 30.0, 0.3, 1.0, 0.05)
    assert not applied and len(got) == 0
    got, _ = gpu_ctx.dynamic_filter(7, q, 0.0, 0.3, 1.0, 0.05)          # radius 0: everything is "outside", all kept
    assert same_points(got, q)
    with pytest.raises(lisreg.LisregError):
        gpu_ctx.dynamic_filter(99, q, 30.0)
    with pytest.raises(lisreg.LisregError):
        gpu_ctx.dynamic_filter(7, q, 30.0, float("nan"), 1.0, 0.05)
    with pytest.raises(lisreg.LisregError):
        gpu_ctx.nearest(7, q, float("nan"))
    # device records in / out, map index over device memory
    rm, rq = lisreg.pack_device_records(m), lisreg.pack_device_records(q)
    dm, dq, dout = lisreg.DeviceArray(rm), lisreg.DeviceArray(rq), lisreg.DeviceArray(np.zeros_like(rq))
    gpu_ctx.map_index_set_device(8, dm.ptr, len(rm))
    n = gpu_ctx.dynamic_filter_device(8, dq.ptr, len(rq), 30.0, 0.3, 1.0, 0.05, dout.ptr)
    want, _ = oracle.dynamic_filter(m, q, 30.0, 0.3, 1.0, 0.05)
    assert n == len(want)
    assert np.array_equal(lisreg.device_to_host(dout.ptr, rq.shape, np.float32)[:n].view(np.uint32), lisreg.pack_device_records(want).view(np.uint32))
    n2 = gpu_ctx.dynamic_filter_device(8, dq.ptr, len(rq), 30.0, 0.3, 1.0, 0.05, dq.ptr)      # in place
    assert n2 == n and np.array_equal(lisreg.device_to_host(dq.ptr, rq.shape, np.float32)[:n], lisreg.device_to_host(dout.ptr, rq.shape, np.float32)[:n])


@pytest.mark.gpu
def test_hip_bbx_and_bounds_match_oracle(oracle, gpu_ctx):
    m, q = _scene(33, n_map=20000)
    assert np.array_equal(gpu_ctx.cloud_bounds(q), oracle.cloud_bounds(q))
    assert np.array_equal(gpu_ctx.cloud_bounds(q[:0]), oracle.cloud_bounds(q[:0]))
    x17 = float(q["x"][17])
    box = np.array([x17, -30.0, -2.0, x17 + 25.0, float(q["y"][400]) + 1e-9, 10.0])     # a double bound between two floats
    for delete in (False, True):
        assert same_points(gpu_ctx.bbx_filter(q, box, delete), oracle.bbx_filter(q, box, delete))
    assert len(gpu_ctx.bbx_filter(q, [1, 1, 1, 0, 0, 0])) == 0           # inverted box keeps nothing
    assert same_points(gpu_ctx.bbx_filter(q, [1, 1, 1, 0, 0, 0], True), q)
    assert len(gpu_ctx.bbx_filter(q[:0], box)) == 0
    # the reference's use (subMapOptmizationNode.cpp:1392-1405): crop the map to the padded intersection with the scan's box
    bq, bm = oracle.cloud_bounds(q), oracle.cloud_bounds(m)
    inter = np.concatenate([np.maximum(bq[:3], bm[:3]) - 2.0, np.minimum(bq[3:], bm[3:]) + 2.0])
    assert same_points(gpu_ctx.bbx_filter(m, inter), oracle.bbx_filter(m, inter))
