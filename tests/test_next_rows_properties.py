"""SURVEY.md §8 f rows at BASELINE-sized inputs (1 M-point map, 128x2048 sweep): size-independent properties, no oracle in
the loop (the CPU restatement would take minutes at these sizes), plus a device-memory leak check of the context life cycle."""
import ctypes as C

import numpy as np
import pytest

f32 = np.float32
pytestmark = pytest.mark.gpu


def _cat(a, b):
    o = np.zeros(len(a) + len(b), a.dtype)
    o[: len(a)], o[len(a):] = a, b
    return o


@pytest.fixture(scope="module")
def big_map():
    from lisreg import synth
    mc, ms = synth.make_submap(1000000, seed=77, labelled=True)
    return _cat(mc, ms)


def test_nearest_of_the_map_itself_is_itself(gpu_ctx, big_map):
    gpu_ctx.map_index_set(20, big_map)
    q = big_map[::3].copy()                                            # 333 k queries: the one-lane-per-query / 4-lane regime
    idx, d2 = gpu_ctx.nearest(20, q)
    assert np.all(d2 == 0.0)
    own = np.arange(0, len(big_map), 3)
    assert np.all(idx <= own)                                           # itself, or an exact duplicate with a smaller index
    from lisreg import synth
    assert np.array_equal(synth.pcl_xyz(big_map[idx]), synth.pcl_xyz(q))
    small = big_map[:5000].copy()                                       # 8 lanes per query
    idx8, d8 = gpu_ctx.nearest(20, small)
    assert np.all(d8 == 0.0) and np.all(idx8 <= np.arange(5000))
    # a rigid shift by s: every nearest distance is at most s (the point's own image is that close), and the capped search agrees
    s = 0.07
    sh = q.copy(); sh["x"] += f32(s)
    _, d2s = gpu_ctx.nearest(20, sh)
    assert d2s.max() <= (s * 1.001) ** 2
    idx_c, d2_c = gpu_ctx.nearest(20, sh, 0.05)
    inside = d2s <= f32(0.05) * f32(0.05)
    assert np.array_equal(idx_c >= 0, inside) and np.array_equal(d2_c[inside], d2s[inside])


def test_dynamic_filter_is_monotone_and_order_preserving(gpu_ctx, big_map):
    from lisreg import synth
    gpu_ctx.map_index_set(20, big_map)
    sc = synth.make_scan(128, 2048, 4242, labelled=True)
    q = _cat(sc["corner"], sc["surf"])
    M = synth.pose_matrix(sc["T_true"])
    w = synth.pcl_xyz(q).astype(np.float64) @ M[:3, :3].T + M[:3, 3] + np.random.default_rng(3).normal(0, 0.2, (len(q), 3))
    q["x"], q["y"], q["z"] = w[:, 0].astype(f32), w[:, 1].astype(f32), w[:, 2].astype(f32)
    q["intensity"] = np.arange(len(q), dtype=f32)                       # a serial number to check the order with
    a, _ = gpu_ctx.dynamic_filter(20, q, 40.0, 0.3, 1.0, 0.05)
    b, _ = gpu_ctx.dynamic_filter(20, q, 40.0, 0.3, 2.0, 0.05)           # wider removal band -> subset
    c, _ = gpu_ctx.dynamic_filter(20, q, 20.0, 0.3, 1.0, 0.05)           # smaller radius -> fewer points examined -> superset
    assert 0 < len(b) <= len(a) <= len(c) < len(q)
    for out in (a, b, c):
        assert np.all(np.diff(out["intensity"]) > 0)                   # input order kept
    sa, sb, sc_ = set(a["intensity"].tolist()), set(b["intensity"].tolist()), set(c["intensity"].tolist())
    assert sb <= sa <= sc_
    _, d2 = gpu_ctx.nearest(20, q)
    r2 = q["x"] * q["x"] + q["y"] * q["y"]
    keep = (r2 > f32(40.0) * f32(40.0)) | ((d2 > f32(0.05) * f32(0.05)) & (d2 < f32(0.3) * f32(0.3))) | (d2 > f32(1.0) * f32(1.0))
    assert np.array_equal(a["intensity"], q["intensity"][keep])         # consistent with the k = 1 distances


def test_voxel_grid_and_crop_at_scale(gpu_ctx, big_map):
    from lisreg import synth
    st, ds = gpu_ctx.voxel_downsample(big_map, 0.4)
    xyz = synth.pcl_xyz(big_map)
    inv = f32(1) / f32(0.4)
    mn = np.floor(xyz.min(0) * inv).astype(np.int64); mx = np.floor(xyz.max(0) * inv).astype(np.int64)
    div = mx - mn + 1
    ijk = (np.floor(xyz * inv) - mn.astype(f32)).astype(np.int64)
    vid = ijk[:, 0] + ijk[:, 1] * div[0] + ijk[:, 2] * div[0] * div[1]
    assert st == 0 and len(ds) == len(np.unique(vid))                   # one output point per occupied PCL voxel
    dxyz = synth.pcl_xyz(ds)
    dj = (np.floor(dxyz * inv) - mn.astype(f32)).astype(np.int64)
    dvid = dj[:, 0] + dj[:, 1] * div[0] + dj[:, 2] * div[0] * div[1]
    assert np.mean(dvid == np.unique(vid)) > 0.999                      # ascending voxel order; a centroid may round onto a face
    st2, ds2 = gpu_ctx.voxel_downsample(ds, 0.4)
    assert len(ds2) <= len(ds) and len(ds2) > 0.95 * len(ds)           # nearly idempotent (centroids on a face can merge)
    b = gpu_ctx.cloud_bounds(big_map)
    assert np.array_equal(b, np.concatenate([xyz.min(0), xyz.max(0)]).astype(np.float64))
    box = np.array([-10.0, -15.0, -1.0, 20.0, 25.0, 3.0])
    inside = np.all((xyz.astype(np.float64) > box[:3]) & (xyz.astype(np.float64) < box[3:]), axis=1)
    kept, dropped = gpu_ctx.bbx_filter(big_map, box), gpu_ctx.bbx_filter(big_map, box, True)
    assert len(kept) == inside.sum() and len(kept) + len(dropped) == len(big_map)
    assert np.array_equal(synth.pcl_xyz(kept), xyz[inside])


def test_icp_round_trip_at_scale(gpu_ctx, big_map):
    """Aligning a rigidly moved subset of the map back onto the map recovers the inverse motion (both ICP variants)."""
    import lisreg
    from lisreg import synth
    gpu_ctx.map_index_set(20, big_map)
    src = big_map[::8].copy()                                           # 125 k points
    M = synth.pose_matrix([0.004, -0.003, 0.01, 0.06, -0.05, 0.02])
    w = synth.pcl_xyz(src).astype(np.float64) @ M[:3, :3].T + M[:3, 3]
    src["x"], src["y"], src["z"] = w[:, 0].astype(f32), w[:, 1].astype(f32), w[:, 2].astype(f32)
    want = np.linalg.inv(M)
    p = lisreg.icp_default_params(0)
    p.max_corr_dist = 1.0; p.transformation_epsilon = 1e-10; p.euclidean_fitness_epsilon = 1e-9; p.max_iters = 40
    r = gpu_ctx.icp_align(20, src, p)
    assert r["converged"] and r["fitness"] < 1e-4
    assert np.abs(r["T"][:3, :3] - want[:3, :3]).max() < 2e-4 and np.abs(r["T"][:3, 3] - want[:3, 3]).max() < 5e-3
    g = gpu_ctx.icp_gn_match(20, src, 25, 1.0, np.eye(4, dtype=f32))
    assert g["steps_applied"] == 25 and g["fitness"] < 1e-4
    assert np.abs(g["T"][:3, :3] - want[:3, :3]).max() < 2e-4 and np.abs(g["T"][:3, 3] - want[:3, 3]).max() < 5e-3


def test_feature_extraction_at_scale_is_permutation_consistent(gpu_ctx):
    """128x2048 sweep: pixel ownership goes to the first point in input order, so reversing the input keeps every pixel
    occupied (same count) and the range image's geometry; the semantic split partitions the cloud."""
    import lisreg
    from lisreg import synth
    c = synth.make_raw_scan(128, 2048, 9100, dup_fraction=0.0)
    p = lisreg.FeatureParams(128, 2048, 1, 0.0, 70.0, 1.0, 0.1)
    a, b = gpu_ctx.extract_features(c, p), gpu_ctx.extract_features(c[::-1].copy(), p)
    assert len(a["deskewed"]) == len(b["deskewed"]) > 200000
    key = lambda r: np.sort(r["ring"].astype(np.int64) * 10 ** 9 + np.round(r["time"].astype(np.float64) * 1e9).astype(np.int64))
    assert np.array_equal(key(a["deskewed"]), key(b["deskewed"]))       # without duplicates per pixel the owners are the same points
    assert len(a["corner"]) <= 128 * 120 and len(a["corner_sharp"]) <= 128 * 24
    assert len(a["corner"]) + len(a["surface"]) <= len(a["deskewed"]) + len(a["corner"])


def test_context_life_cycle_does_not_leak_device_memory():
    import lisreg
    from lisreg import synth
    hip = lisreg.hip_runtime()
    def free_bytes():
        fr, tot = C.c_size_t(), C.c_size_t()
        assert hip.hipMemGetInfo(C.byref(fr), C.byref(tot)) == 0
        return fr.value
    case = synth.make_case(h=16, w=450, m_points=20000, scan_seed=1500)
    def cycle():
        ctx = lisreg.Context(0)
        ctx.set_target(case["tgt_corner"], case["tgt_surf"])
        ctx.align(case["src_corner"], case["src_surf"], case["T_init"], lisreg.default_params(1))
        ctx.map_index_set(0, case["tgt_surf"])
        ctx.nearest(0, case["src_surf"])
        ctx.icp_align(0, case["src_surf"], lisreg.icp_default_params(0))
        ctx.voxel_downsample(case["tgt_surf"], 0.4)
        ctx.close() if hasattr(ctx, "close") else None
        del ctx
    cycle()                                                              # warm the runtime's own pools
    before = free_bytes()
    for _ in range(10):
        cycle()
    import gc; gc.collect()
    assert before - free_bytes() < 8 * 2 ** 20                          # nothing accumulates over 10 life cycles
