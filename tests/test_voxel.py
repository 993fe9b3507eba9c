"""pcl::VoxelGrid replacement and transformPointCloud.

CPU: the oracle's restatement against an independent numpy computation of PCL's voxel index / centroid / majority label.
GPU: liblisreg against the oracle — voxel membership, output ORDER (ascending PCL voxel index) and labels exactly;
centroids bit for bit, because both sum every voxel's points sequentially in input order (PCL's own std::sort is
unstable, so the reference fixes no order)."""
import numpy as np
import pytest

f32 = np.float32


def _cloud(seed, n_scan=(32, 900), labelled=True):
    from lisreg import synth
    sc = synth.make_scan(n_scan[0], n_scan[1], seed, labelled=labelled)
    c = sc["surf"].copy()
    rng = np.random.default_rng(seed)
    c["intensity"] = rng.uniform(0, 255, len(c)).astype(f32)
    if labelled:                                  # mix labels inside voxels so the majority vote matters
        flip = rng.random(len(c)) < 0.3
        c["label"][flip] = rng.integers(0, 20, int(flip.sum())).astype(np.uint16)
    return c


def _numpy_voxel(c, leaf):
    from lisreg import synth
    xyz = synth.pcl_xyz(c)
    inv = f32(1) / f32(leaf)
    mn = np.floor(xyz.min(0) * inv).astype(np.int64); mx = np.floor(xyz.max(0) * inv).astype(np.int64)
    div = mx - mn + 1
    ijk = (np.floor(xyz * inv) - mn.astype(f32)).astype(np.int64)
    idx = ijk[:, 0] + ijk[:, 1] * div[0] + ijk[:, 2] * div[0] * div[1]
    u, inverse, cnt = np.unique(idx, return_inverse=True, return_counts=True)
    cen = np.zeros((len(u), 4)); np.add.at(cen, inverse, np.concatenate([xyz, c["intensity"][:, None]], 1).astype(np.float64))
    cen /= cnt[:, None]
    lab = np.zeros(len(u), np.uint16)
    order = np.argsort(idx, kind="stable")
    starts = np.concatenate([[0], np.cumsum(cnt)])
    for v in range(len(u)):
        l = c["label"][order[starts[v]:starts[v + 1]]]
        vals, k = np.unique(l, return_counts=True)
        lab[v] = vals[np.argmax(k)]              # np.unique sorts ascending: first max = smallest label on ties
    return cen, lab


@pytest.mark.parametrize("leaf", [0.2, 0.4, 1.0])
def test_oracle_voxel_grid_matches_numpy(oracle, leaf):
    from lisreg import synth
    c = _cloud(11)
    rc, ds = oracle.voxel_grid(c, leaf)
    cen, lab = _numpy_voxel(c, leaf)
    assert rc == 0 and len(ds) == len(cen)
    got = np.concatenate([synth.pcl_xyz(ds), ds["intensity"][:, None]], 1)
    assert np.abs(got[:, :3] - cen[:, :3]).max() < 2e-5 and np.abs(got[:, 3] - cen[:, 3]).max() < 2e-3
    assert np.array_equal(ds["label"], lab)


def test_oracle_voxel_grid_edges(oracle):
    from lisreg import synth
    c = _cloud(12)
    rc, ds = oracle.voxel_grid(c[:0], 0.4)
    assert rc == 0 and len(ds) == 0
    rc, ds = oracle.voxel_grid(c[:1], 0.4)
    assert rc == 0 and len(ds) == 1 and ds[0]["x"] == c[0]["x"] and ds[0]["label"] == c[0]["label"]
    rc, ds = oracle.voxel_grid(c, 0.001)          # (80/0.001)^2 * ... overflows int32 -> PCL copies the input
    assert rc == 3 and len(ds) == len(c) and np.array_equal(ds["x"], c["x"])
    rc, ds = oracle.voxel_grid(c, 500.0)          # PCL's grid is anchored at the ORIGIN: a cloud straddling it -> 2 per axis
    cen, lab = _numpy_voxel(c, 500.0)
    assert rc == 0 and len(ds) == len(cen) == 8 and np.array_equal(ds["label"], lab)


def test_oracle_transform_cloud(oracle):
    from lisreg import synth
    c = _cloud(13)
    T = np.array([0.01, -0.02, 0.7, 3.0, -4.0, 0.5], f32)
    out = oracle.transform_cloud(c, T)
    M = synth.pose_matrix(T)
    ref = synth.pcl_xyz(c).astype(np.float64) @ M[:3, :3].T + M[:3, 3]
    assert np.abs(synth.pcl_xyz(out) - ref).max() < 2e-5
    assert np.array_equal(out["intensity"], c["intensity"]) and np.array_equal(out["label"], c["label"])


@pytest.mark.gpu
@pytest.mark.parametrize("leaf,labelled,seed", [(0.2, True, 21), (0.4, True, 22), (0.4, False, 23), (0.05, True, 24), (2.0, True, 25)])
def test_hip_voxel_grid_matches_oracle(oracle, gpu_ctx, leaf, labelled, seed):
    c = _cloud(seed, labelled=labelled)
    if not labelled:
        from lisreg import synth
        c = synth.to_pcl(synth.pcl_xyz(c), None, c["intensity"])     # still 32-B structs, label field zero
    rc_o, do = oracle.voxel_grid(c, leaf, fmt=1 if labelled else 0)
    rc_g, dg = gpu_ctx.voxel_downsample(c, leaf)
    assert rc_g == rc_o == 0 and len(dg) == len(do)
    for f in ("x", "y", "z", "intensity"):
        assert np.array_equal(dg[f], do[f]), f                       # same order, same float sums
    assert np.array_equal(dg["label"], do["label"])


@pytest.mark.gpu
def test_hip_voxel_grid_edges_and_device_format(oracle, gpu_ctx):
    import lisreg
    from lisreg import synth
    c = _cloud(26)
    rc, d = gpu_ctx.voxel_downsample(c[:0], 0.4)
    assert rc == 0 and len(d) == 0
    rc, d = gpu_ctx.voxel_downsample(c, 0.001)
    assert rc == lisreg.LEAF_TOO_SMALL and len(d) == len(c) and np.array_equal(d["x"], c["x"])
    rc, d = gpu_ctx.voxel_downsample(c, 500.0)
    ro, do = oracle.voxel_grid(c, 500.0)
    assert rc == 0 and len(d) == len(do) == 8                        # grid anchored at the origin: 2 voxels per axis
    assert np.array_equal(d["x"], do["x"]) and np.array_equal(d["label"], do["label"])
    # device-resident records in, device-resident records out: xyz centroids + majority label in the payload
    rec = lisreg.pack_device_records(c)
    din, dout = lisreg.DeviceArray(rec), lisreg.DeviceArray(np.zeros_like(rec))
    rc, n_out = gpu_ctx.voxel_downsample_device(din.ptr, len(c), 0.4, dout.ptr, len(c))
    ro, do = oracle.voxel_grid(c, 0.4)
    got = lisreg.device_to_host(dout.ptr, (len(c), 4))[:n_out]
    assert rc == 0 and n_out == len(do)
    assert np.array_equal(got[:, 0], do["x"]) and np.array_equal(got[:, 1], do["y"]) and np.array_equal(got[:, 2], do["z"])
    assert np.array_equal(got[:, 3].copy().view(np.uint32) & 0xffff, do["label"].astype(np.uint32))
    # too small an output buffer: nothing written, the needed size reported
    with pytest.raises(lisreg.LisregError):
        gpu_ctx.voxel_downsample_device(din.ptr, len(c), 0.4, dout.ptr, 10)


@pytest.mark.gpu
def test_hip_transform_cloud_matches_oracle(oracle, gpu_ctx):
    import lisreg
    c = _cloud(27)
    T = np.array([0.02, -0.01, -1.3, 12.0, -7.5, 0.25], f32)
    o = oracle.transform_cloud(c, T)
    g = gpu_ctx.transform_cloud(c, T)
    from lisreg import synth
    assert np.abs(synth.pcl_xyz(g) - synth.pcl_xyz(o)).max() <= 8e-6        # FMA contraction only
    assert np.array_equal(g["intensity"], c["intensity"]) and np.array_equal(g["label"], c["label"])
    rec = lisreg.pack_device_records(c)
    d = lisreg.DeviceArray(rec)
    gpu_ctx.transform_cloud_device(d.ptr, len(c), T, d.ptr)                  # in place
    got = lisreg.device_to_host(d.ptr, (len(c), 4))
    assert np.array_equal(got[:, :3], synth.pcl_xyz(g)) and np.array_equal(got[:, 3].view(np.uint32), rec[:, 3].view(np.uint32))


@pytest.mark.gpu
def test_downsample_then_register_pipeline(oracle, gpu_ctx):
    """The reference's per-frame order (odomEstimationNode.cpp:185-207, 260-281, 596): voxel-grid the local map and the
    incoming features, then register.  HIP pipeline vs oracle pipeline."""
    import lisreg
    from helpers import pose_err
    from lisreg import synth
    case = synth.make_case(h=32, w=900, m_points=60000, scan_seed=1500)
    out_g, out_o = {}, {}
    for key, leaf in (("tgt_corner", 0.2), ("tgt_surf", 0.4), ("src_corner", 0.2), ("src_surf", 0.4)):
        rg, out_g[key] = gpu_ctx.voxel_downsample(case[key], leaf)
        ro, out_o[key] = oracle.voxel_grid(case[key], leaf)
        assert rg == ro == 0 and len(out_g[key]) == len(out_o[key])
    p_o = oracle.default_params(1)
    To, so, _ = oracle.align(out_o["tgt_corner"], out_o["tgt_surf"], out_o["src_corner"], out_o["src_surf"], case["T_init"], p_o)
    gpu_ctx.set_target(out_g["tgt_corner"], out_g["tgt_surf"])
    Tg, sg, _ = gpu_ctx.align(out_g["src_corner"], out_g["src_surf"], case["T_init"], lisreg.default_params(1))
    assert sg["status"] == so["status"] == 0
    assert max(pose_err(Tg, To)) <= 1e-3


@pytest.mark.gpu
def test_multi_cloud_grid_equals_single_calls_bitwise(gpu_ctx):
    """lisreg_voxel_downsample_multi: K device clouds through one sort and one centroid launch — every cloud's output (membership, order,
    centroids, label votes / intensity averages) identical to its own lisreg_voxel_downsample call; empty clouds, a single live cloud,
    a cloud whose leaf is too small for it (falls back to single calls) and in-place use."""
    import lisreg
    from lisreg import synth
    rng = np.random.default_rng(5)
    tc, ts = synth.make_submap(120000, 77, labelled=True)
    clouds = [ts[:50000], tc, ts[50000:90000], ts[:0], ts[90000:]]
    leafs = [0.4, 0.05, 0.2, 0.3, 0.6]
    recs = [lisreg.pack_device_records(c) for c in clouds]
    for intensity in (False, True):
        bufs_in = [lisreg.DeviceArray(r if len(r) else np.zeros((1, 4), np.float32)) for r in recs]
        singles = []
        for b, r, lf in zip(bufs_in, recs, leafs):
            o = lisreg.DeviceArray(np.zeros((max(len(r), 1), 4), np.float32))
            n = gpu_ctx.voxel_downsample_device(b.ptr, len(r), lf, o.ptr, max(len(r), 1), intensity=intensity)[1] if len(r) else 0
            singles.append(o.download(n))
            o.free()
        outs = [lisreg.DeviceArray(np.zeros((max(len(r), 1), 4), np.float32)) for r in recs]
        counts = gpu_ctx.voxel_downsample_multi_device([b.ptr for b in bufs_in], [len(r) for r in recs], leafs, [o.ptr for o in outs],
                                                       [max(len(r), 1) for r in recs], intensity=intensity)
        assert counts == [len(s) for s in singles] and counts[3] == 0
        for o, s, cnt in zip(outs, singles, counts):
            assert np.array_equal(o.download(cnt).view(np.uint32), s.view(np.uint32))
        # in place (the local map's five grids), and one live cloud only
        counts2 = gpu_ctx.voxel_downsample_multi_device([b.ptr for b in bufs_in], [len(r) for r in recs], leafs, [b.ptr for b in bufs_in],
                                                        [max(len(r), 1) for r in recs], intensity=intensity)
        assert counts2 == counts
        for b, s, cnt in zip(bufs_in, singles, counts):
            assert np.array_equal(b.download(cnt).view(np.uint32), s.view(np.uint32))
        for b in bufs_in + outs:
            b.free()
    # a leaf far too small for its cloud ("Leaf size is too small for the input dataset": output = input) -> per-cloud path, same results
    a = lisreg.DeviceArray(recs[0]); b = lisreg.DeviceArray(recs[1])
    oa = lisreg.DeviceArray(np.zeros_like(recs[0])); ob = lisreg.DeviceArray(np.zeros_like(recs[1]))
    cnt = gpu_ctx.voxel_downsample_multi_device([a.ptr, b.ptr], [len(recs[0]), len(recs[1])], [1e-4, 0.05], [oa.ptr, ob.ptr], [len(recs[0]), len(recs[1])])
    assert cnt[0] == len(recs[0]) and np.array_equal(oa.download(cnt[0]), recs[0])
    for x in (a, b, oa, ob):
        x.free()


@pytest.mark.gpu
@pytest.mark.parametrize("labelled", [True, False])
@pytest.mark.parametrize("seed", [31, 32, 33])
def test_hip_voxel_grid_crowded_voxels(oracle, gpu_ctx, labelled, seed):
    """Voxels that hold tens to thousands of points (the cells next to the sensor, a key-frame ring's overlap): the wavefront-per-voxel
    path sums them 64 at a time through LDS, in input order — centroid, intensity average and label vote equal the oracle's bit for bit,
    for voxel populations around every chunk boundary (24 / 25, 63 / 64 / 65, 128, several thousand)."""
    from lisreg import synth
    rng = np.random.default_rng(seed)
    pops = [1, 2, 23, 24, 25, 26, 63, 64, 65, 127, 128, 129, 500, 4097] + [int(v) for v in rng.integers(1, 300, 40)]
    xyz, lab = [], []
    for k, p in enumerate(pops):                       # voxel k: p points inside one 0.5 m cell of a 3-D lattice, 14 possible labels
        cell = np.array([k % 7, (k // 7) % 7, k // 49], np.float64) * 0.5 + 10.0
        xyz.append(cell + rng.uniform(0.01, 0.49, (p, 3)))
        lab.append(rng.integers(0, 14, p) if k % 3 else np.full(p, 7))
    xyz = np.concatenate(xyz).astype(np.float32); lab = np.concatenate(lab).astype(np.uint16)
    perm = rng.permutation(len(xyz))                   # input order is not voxel order
    cloud = synth.to_pcl(xyz[perm], lab[perm])
    cloud["intensity"] = rng.uniform(0, 255, len(cloud)).astype(np.float32)
    st_o, want = oracle.voxel_grid(cloud, 0.5, fmt=1 if labelled else 0)
    st_g, got = gpu_ctx.voxel_downsample(cloud, 0.5) if labelled else gpu_ctx.voxel_downsample(_as_xyzi(cloud), 0.5)
    assert st_o == st_g == 0 and len(got) == len(want) == len(pops)
    for f in ("x", "y", "z", "intensity") + (("label",) if labelled else ()):
        assert np.array_equal(got[f], want[f]), f


def _as_xyzi(cloud):
    """the same points as a PointXYZI array (no label field): the grid then averages the intensity and votes nothing"""
    out = np.zeros(len(cloud), np.dtype({"names": ["x", "y", "z", "intensity"], "formats": ["<f4"] * 4, "offsets": [0, 4, 8, 16], "itemsize": 32}))
    for f in ("x", "y", "z", "intensity"):
        out[f] = cloud[f]
    return out
