"""range-image projection + LOAM feature extraction (the producer of cloud_info).

CPU: the C oracle against an independent pure-Python mirror written separately from the reference text.
GPU: liblisreg against the oracle — all five index lists (deskewed, corner, surface, corner_sharp, surface_sharp)
identical, element for element and in the reference's order: this is integer / index work, so the bar is exact."""
import numpy as np
import pytest

NAMES = ("deskewed", "corner", "surface", "corner_sharp", "surface_sharp")


def same_points(got, want):
    """Field-by-field equality of two PointXYZIRT arrays (numpy fancy indexing does not preserve struct padding)."""
    return len(got) == len(want) and all(np.array_equal(got[f], want[f]) for f in want.dtype.names)


def _pd(p):
    return {f: getattr(p, f) for f, _ in p._fields_}


@pytest.mark.parametrize("h,w,rate,seed,shuffle", [(8, 240, 1, 1, False), (16, 450, 2, 2, False), (16, 450, 1, 3, True)])
def test_oracle_matches_python_mirror(oracle, h, w, rate, seed, shuffle):
    import lisreg_numpy as ln
    from lisreg import synth
    c = synth.make_raw_scan(h, w, 7000 + seed, shuffle=shuffle)
    p = oracle.FeatureParams(h, w, rate, 0.0, 70.0, 1.0, 0.1)
    a = oracle.extract_features(c, p)
    b = ln.extract_features(c["x"], c["y"], c["z"], c["ring"], _pd(p))
    for k in NAMES:
        assert np.array_equal(a[k], b[k]), k
    assert len(a["corner"]) > 0 and len(a["surface_sharp"]) > 0 and len(a["deskewed"]) <= h * w


def test_oracle_feature_invariants(oracle):
    from lisreg import synth
    c = synth.make_raw_scan(64, 1800, 7100)
    p = oracle.default_feature_params()
    r = oracle.extract_features(c, p)
    assert len(np.unique(r["deskewed"])) == len(r["deskewed"])                   # one point per pixel
    assert np.all(c["ring"][r["deskewed"]] % p.downsample_rate == 0)             # dropped rows stay empty
    assert np.all(np.diff(c["ring"][r["deskewed"]].astype(int)) >= 0)            # row-major order
    assert set(r["corner_sharp"]) <= set(r["corner"]) and set(r["surface_sharp"]) <= set(r["surface"])
    assert not (set(r["corner"]) & set(r["surface"]))                           # label 1 never enters the surface cloud
    # per ring at most 6*20 corners, 6*4 sharp corners, 6*10 sharp surfaces
    for name, cap in (("corner", 120), ("corner_sharp", 24), ("surface_sharp", 60)):
        assert np.bincount(c["ring"][r[name]], minlength=64).max() <= cap
    # first point in input order wins a pixel: shuffling the input changes owners, not the pixel count
    c2 = synth.make_raw_scan(64, 1800, 7100, shuffle=True)
    assert len(oracle.extract_features(c2, p)["deskewed"]) == len(r["deskewed"])
    # range limits
    p.max_range = 20.0
    r2 = oracle.extract_features(c, p)
    rng = np.sqrt(c["x"] ** 2 + c["y"] ** 2 + c["z"] ** 2)
    assert len(r2["deskewed"]) < len(r["deskewed"]) and rng[r2["deskewed"]].max() <= 20.0


@pytest.mark.gpu
@pytest.mark.parametrize("h,w,rate,seed,shuffle", [(16, 450, 1, 11, False), (64, 1800, 2, 12, False), (64, 1800, 1, 13, True),
                                                   (128, 2048, 1, 14, False), (32, 1024, 4, 15, True)])
def test_hip_features_match_oracle(oracle, gpu_ctx, h, w, rate, seed, shuffle):
    import lisreg
    from lisreg import synth
    c = synth.make_raw_scan(h, w, 7200 + seed, shuffle=shuffle)
    po = oracle.FeatureParams(h, w, rate, 0.0, 70.0, 1.0, 0.1)
    pg = lisreg.FeatureParams(h, w, rate, 0.0, 70.0, 1.0, 0.1)
    ro = oracle.extract_features(c, po)
    rg = gpu_ctx.extract_features(c, pg)
    for k in NAMES:
        assert len(rg[k]) == len(ro[k]), k
        assert same_points(rg[k], c[ro[k]]), k                                # the same points, in the same order, all fields


@pytest.mark.gpu
def test_hip_features_edges_and_device_format(oracle, gpu_ctx):
    import lisreg
    from lisreg import synth
    c = synth.make_raw_scan(16, 450, 7300)
    pg = lisreg.FeatureParams(16, 450, 1, 0.0, 70.0, 1.0, 0.1)
    po = oracle.FeatureParams(16, 450, 1, 0.0, 70.0, 1.0, 0.1)
    r = gpu_ctx.extract_features(c[:0], pg)
    assert all(len(r[k]) == 0 for k in NAMES)
    few = c[:40]                                                              # fewer points than the 11-tap stencil needs per ring
    ro, rg = oracle.extract_features(few, po), gpu_ctx.extract_features(few, pg)
    for k in NAMES:
        assert same_points(rg[k], few[ro[k]]), k
    far = c.copy(); far["x"] += 500.0                                         # everything beyond lidarMaxRange
    assert all(len(v) == 0 for v in gpu_ctx.extract_features(far, pg).values())
    with pytest.raises(lisreg.LisregError):
        gpu_ctx.extract_features(c, lisreg.FeatureParams(16, 8000, 1, 0.0, 70.0, 1.0, 0.1))
    # device records in (ring in the payload), device records out
    rec = np.zeros((len(c), 4), np.float32)
    rec[:, 0], rec[:, 1], rec[:, 2] = c["x"], c["y"], c["z"]
    rec[:, 3] = c["ring"].astype(np.uint32).view(np.float32)
    din = lisreg.DeviceArray(rec)
    cap = 16 * 450
    outs = {k: lisreg.DeviceArray(np.zeros((cap, 4), np.float32)) for k in NAMES}
    counts = gpu_ctx.extract_features_device(din.ptr, len(c), pg, {k: v.ptr for k, v in outs.items()}, cap)
    ro = oracle.extract_features(c, po)
    for k in NAMES:
        assert counts[k] == len(ro[k]), k
        got = lisreg.device_to_host(outs[k].ptr, (cap, 4))[: counts[k]]
        assert got.tobytes() == rec[ro[k]].tobytes(), k


@pytest.mark.gpu
def test_full_front_end_pipeline_matches_oracle(oracle, gpu_ctx):
    """The reference's per-frame chain: laserProcessing (project + extract features) -> odomEstimation (voxel-grid the
    features, register against the local map).  HIP chain vs oracle chain: identical features, identical down-sampled
    clouds, pose within the 1e-3 bar."""
    import lisreg
    from helpers import pose_err
    from lisreg import synth
    h, w = 64, 1800
    sweep = synth.make_raw_scan(h, w, 1000)
    sc_truth = synth.make_scan(h, w, 1000)["T_true"]
    po = oracle.FeatureParams(h, w, 1, 0.0, 70.0, 1.0, 0.1); pg = lisreg.FeatureParams(h, w, 1, 0.0, 70.0, 1.0, 0.1)
    fo, fg = oracle.extract_features(sweep, po), gpu_ctx.extract_features(sweep, pg)

    def as_xyzi(c):                                   # PointXYZIRT -> PointXYZI (what pcl::fromROSMsg hands the odometry node)
        return synth.to_pcl(np.stack([c["x"], c["y"], c["z"]], 1), None, c["intensity"])
    clouds_o = {k: as_xyzi(sweep[fo[k]]) for k in ("corner", "surface")}
    clouds_g = {k: as_xyzi(fg[k]) for k in ("corner", "surface")}
    for k in clouds_o:
        assert same_points(clouds_g[k], clouds_o[k]), k
    tc, ts = synth.make_submap(200000, 42)
    ds_o, ds_g = {}, {}
    for name, cloud_o, cloud_g, leaf in (("tc", tc, tc, 0.2), ("ts", ts, ts, 0.4), ("sc", clouds_o["corner"], clouds_g["corner"], 0.2),
                                         ("ss", clouds_o["surface"], clouds_g["surface"], 0.4)):
        ro, ds_o[name] = oracle.voxel_grid(cloud_o, leaf, fmt=0)
        rg, ds_g[name] = gpu_ctx.voxel_downsample(cloud_g, leaf)
        assert ro == rg == 0 and same_points(ds_g[name], ds_o[name]), name
    T0 = synth.perturb_pose(sc_truth, np.random.default_rng(3))
    To, so, _ = oracle.align(ds_o["tc"], ds_o["ts"], ds_o["sc"], ds_o["ss"], T0, oracle.default_params(1), fmt=0)
    gpu_ctx.set_target(ds_g["tc"], ds_g["ts"])
    Tg, sg, _ = gpu_ctx.align(ds_g["sc"], ds_g["ss"], T0, lisreg.default_params(1))
    assert so["status"] == sg["status"] == 0 and len(ds_g["ss"]) > 2000
    assert max(pose_err(Tg, To)) <= 1e-3 and max(pose_err(Tg, sc_truth)) < 3e-2


def _labelled_cloud(seed, n=50000):
    from lisreg import synth
    rng = np.random.default_rng(seed)
    c = synth.to_pcl(rng.uniform(-40, 40, (n, 3)).astype(np.float32), rng.integers(0, 20, n).astype(np.uint16),
                     rng.uniform(0, 255, n).astype(np.float32))
    return c


def test_oracle_semantic_split_matches_label_yaml(oracle):
    """categoryMapping (semanticFusionNode.cpp:173-189) with config/label.yaml:177-196."""
    c = _labelled_cloud(1)
    parts = oracle.semantic_split(c)
    groups = [{1, 2, 3, 4, 5, 6, 7, 8}, {9, 10, 11}, {13, 14}, {16, 18, 19}, {0, 12, 15, 17}]   # dynamic ground building pole outlier
    assert sum(len(p) for p in parts) == len(c)
    for part, g in zip(parts, groups):
        assert set(np.unique(part["label"]).tolist()) <= g
        want = c[np.isin(c["label"], list(g))]                                 # order-preserving
        assert same_points(part, want)


@pytest.mark.gpu
def test_hip_semantic_split_matches_oracle(oracle, gpu_ctx):
    import lisreg
    c = _labelled_cloud(2, 120000)
    po, pg = oracle.semantic_split(c), gpu_ctx.semantic_split(c)
    assert [len(p) for p in pg] == [len(p) for p in po]
    for a, b in zip(pg, po):
        assert same_points(a, b)
    custom = [81] * 32                                                          # everything is a pole
    pg = gpu_ctx.semantic_split(c, custom)
    assert [len(p) for p in pg] == [0, 0, 0, len(c), 0] and same_points(pg[3], c)
    assert [len(p) for p in gpu_ctx.semantic_split(c[:0])] == [0] * 5
    # the split feeds copies #2/#3: edge set = pole, planar set = dynamic + building + ground (subMapOptmizationNode.cpp:856-893)
    pg = gpu_ctx.semantic_split(c)
    assert len(pg[3]) + len(pg[0]) + len(pg[2]) + len(pg[1]) + len(pg[4]) == len(c)


# ---------------------------------------------------------------- IMU de-skew (deskewPoint / findRotation)
def _imu_tables(seed, t0=100.0, n=70, rate=500.0):
    """integrated IMU rotation like imuDeskewInfo builds it: first entry zero, 500 Hz, a smooth yaw-dominant motion"""
    rng = np.random.default_rng(seed)
    t = t0 - 0.01 + np.arange(n) / rate
    w = np.stack([0.05 * np.sin(6 * (t - t0)), 0.03 * np.cos(4 * (t - t0)), 0.6 + 0.2 * np.sin(3 * (t - t0))], 1) + rng.normal(0, 0.01, (n, 3))
    rot = np.zeros((n, 3))
    rot[1:] = np.cumsum(w[1:] * np.diff(t)[:, None], 0)
    return t, rot


def _numpy_deskew(c, idx, t, rot, t0):
    """double-precision mirror: rotation interpolated at the point time, R_start^-1 R_point applied to the point"""
    from lisreg import synth
    def R_at(pt):
        front = 0
        while front < len(t) - 1 and not (pt < t[front]):
            front += 1
        if pt > t[front] or front == 0:
            r = rot[front]
        else:
            b = front - 1
            rf = (pt - t[b]) / (t[front] - t[b]); rb = (t[front] - pt) / (t[front] - t[b])
            r = rot[front] * rf + rot[b] * rb
        return synth.pose_matrix([r[0], r[1], r[2], 0, 0, 0])[:3, :3]
    first = idx.min()
    Rsi = np.linalg.inv(R_at(t0 + float(c["time"][first])))
    xyz = synth.pcl_xyz(c).astype(np.float64)
    return np.stack([Rsi @ R_at(t0 + float(c["time"][i])) @ xyz[i] for i in idx])


def test_oracle_deskew_matches_numpy(oracle):
    from lisreg import synth
    c = synth.make_raw_scan(16, 450, 7400)
    p = oracle.FeatureParams(16, 450, 1, 0.0, 70.0, 1.0, 0.1)
    idx = oracle.extract_features(c, p)["deskewed"]
    t, rot = _imu_tables(1)
    dk = oracle.make_deskew(t, rot[:, 0], rot[:, 1], rot[:, 2], 100.0)
    got = oracle.deskew_points(c, dk, idx)
    want = _numpy_deskew(c, idx, t, rot, 100.0)
    assert np.abs(got - want).max() < 5e-5                                   # float rotation algebra at <= 70 m
    moved = np.linalg.norm(got - synth.pcl_xyz(c)[idx], axis=1)
    assert moved.max() > 0.5 and moved[np.argmin(idx)] < 1e-5                # the sweep turns ~3 deg; the first point stays put
    off = oracle.make_deskew(t, rot[:, 0], rot[:, 1], rot[:, 2], 100.0, enabled=False)
    assert np.array_equal(oracle.deskew_points(c, off, idx), synth.pcl_xyz(c)[idx])
    # a point time beyond the last IMU sample takes the last rotation (findRotation :382-387)
    late = oracle.make_deskew(t[:10], rot[:10, 0], rot[:10, 1], rot[:10, 2], 100.0)
    g2 = oracle.deskew_points(c, late, idx)
    assert np.isfinite(g2).all()


@pytest.mark.gpu
@pytest.mark.parametrize("h,w,rate,seed,shuffle", [(16, 450, 1, 21, False), (64, 1800, 2, 22, False), (32, 1024, 1, 23, True)])
def test_hip_deskew_matches_oracle(oracle, gpu_ctx, h, w, rate, seed, shuffle):
    import lisreg
    from lisreg import synth
    c = synth.make_raw_scan(h, w, 7500 + seed, shuffle=shuffle)
    po = oracle.FeatureParams(h, w, rate, 0.0, 70.0, 1.0, 0.1)
    pg = lisreg.FeatureParams(h, w, rate, 0.0, 70.0, 1.0, 0.1)
    t, rot = _imu_tables(seed)
    dko = oracle.make_deskew(t, rot[:, 0], rot[:, 1], rot[:, 2], 100.0)
    dkg = lisreg.make_deskew(t, rot[:, 0], rot[:, 1], rot[:, 2], 100.0)
    ro = oracle.extract_features(c, po)
    rg = gpu_ctx.extract_features(c, pg, dkg)
    plain = gpu_ctx.extract_features(c, pg)
    xyz_all = {int(i): v for i, v in zip(ro["deskewed"], oracle.deskew_points(c, dko, ro["deskewed"]))}
    for k in NAMES:
        assert len(rg[k]) == len(ro[k]), k
        want = np.stack([xyz_all[int(i)] for i in ro[k]]) if len(ro[k]) else np.zeros((0, 3), np.float32)
        assert np.array_equal(synth.pcl_xyz(rg[k]), want), k                   # coordinates bit for bit
        for f in ("intensity", "ring", "time"):                                # everything else is the raw point's
            assert np.array_equal(rg[k][f], c[ro[k]][f]), (k, f)
        assert len(plain[k]) == len(rg[k])                                     # the selection does not depend on the de-skew
    assert np.abs(synth.pcl_xyz(rg["deskewed"]) - synth.pcl_xyz(plain["deskewed"])).max() > 0.5
    off = lisreg.make_deskew(t, rot[:, 0], rot[:, 1], rot[:, 2], 100.0, enabled=False)
    r0 = gpu_ctx.extract_features(c, pg, off)
    assert all(same_points(r0[k], plain[k]) for k in NAMES)
    # device records: time as a separate device array
    rec = np.zeros((len(c), 4), np.float32)
    rec[:, 0], rec[:, 1], rec[:, 2] = c["x"], c["y"], c["z"]
    rec[:, 3] = c["ring"].astype(np.uint32).view(np.float32)
    din, dt = lisreg.DeviceArray(rec), lisreg.DeviceArray(np.ascontiguousarray(c["time"]))
    cap = h * w
    outs = {k: lisreg.DeviceArray(np.zeros((cap, 4), np.float32)) for k in NAMES}
    dkd = lisreg.make_deskew(t, rot[:, 0], rot[:, 1], rot[:, 2], 100.0, time_device_ptr=dt.ptr)
    nd = gpu_ctx.extract_features_device(din.ptr, len(c), pg, {k: v.ptr for k, v in outs.items()}, cap, dkd)
    for k in NAMES:
        assert nd[k] == len(rg[k])
        got = lisreg.device_to_host(outs[k].ptr, (cap, 4), np.float32)[: nd[k]]
        assert np.array_equal(got[:, :3], synth.pcl_xyz(rg[k])), k


def _records(c):
    rec = np.zeros((len(c), 4), np.float32)
    rec[:, 0], rec[:, 1], rec[:, 2] = c["x"], c["y"], c["z"]
    rec[:, 3] = c["ring"].astype(np.uint32).view(np.float32)
    return rec


@pytest.mark.gpu
@pytest.mark.parametrize("h,w,rate", [(16, 450, 1), (64, 1800, 2)])
def test_batched_extraction_equals_single_calls_and_oracle(oracle, gpu_ctx, h, w, rate):
    """lisreg_extract_features_batch stacks the sweeps into one range image (grid = sweeps x rings).  Every sweep's five clouds
    must be exactly what a single call — and the oracle — give for that sweep alone: the flat loops of the reference stop 5 / 6
    entries short of the CLOUD's ends, which in the stack are the sweep's own ends.  Includes an empty sweep, a sweep too small for
    the stencil, shuffled input and a sweep whose last ring is empty."""
    import lisreg
    from lisreg import synth
    sweeps = [synth.make_raw_scan(h, w, 7400 + k, shuffle=(k == 2)) for k in range(6)]
    sweeps[1] = sweeps[1][:0]
    sweeps[3] = sweeps[3][:30]
    sweeps[4] = sweeps[4][sweeps[4]["ring"] < h - 2]
    sweeps.append(synth.make_raw_scan(h, w, 7410))
    sweeps.append(synth.make_raw_scan(h, w, 7411))
    pg = lisreg.FeatureParams(h, w, rate, 0.0, 70.0, 1.0, 0.1)
    po = oracle.FeatureParams(h, w, rate, 0.0, 70.0, 1.0, 0.1)
    recs = [_records(c) for c in sweeps]
    dins = [lisreg.DeviceArray(r) if len(r) else None for r in recs]
    cap = h * w
    outs = [{k: lisreg.DeviceArray(np.zeros((cap, 4), np.float32)) for k in NAMES} for _ in sweeps]
    counts = gpu_ctx.extract_features_batch_device([d.ptr if d else 0 for d in dins], [len(r) for r in recs], pg,
                                                   [{k: v.ptr for k, v in o.items()} for o in outs], cap)
    one = {k: lisreg.DeviceArray(np.zeros((cap, 4), np.float32)) for k in NAMES}
    for s, c in enumerate(sweeps):
        ro = oracle.extract_features(c, po)
        single = gpu_ctx.extract_features_device(dins[s].ptr if dins[s] else 0, len(c), pg, {k: v.ptr for k, v in one.items()}, cap)
        for k in NAMES:
            assert counts[s][k] == single[k] == len(ro[k]), (s, k, counts[s][k], single[k], len(ro[k]))
            got = lisreg.device_to_host(outs[s][k].ptr, (cap, 4))[: counts[s][k]]
            assert got.tobytes() == recs[s][ro[k]].tobytes(), (s, k)
    assert sum(c["corner"] for c in counts) > 0
