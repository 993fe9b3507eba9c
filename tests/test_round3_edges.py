"""Edge cases of the launch sequences that were fused in round 3 (one partition for categoryMapping, K clouds per voxel-grid call, the
class crops of localmap_extract in one sequence): empty and one-sided inputs, labels outside the table, against the CPU restatement."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _labelled(n, labels, seed=0):
    from lisreg import synth
    rng = np.random.default_rng(seed)
    xyz = rng.uniform(-20, 20, (n, 3)).astype(np.float32)
    lab = np.asarray(labels, np.uint16) if np.ndim(labels) else np.full(n, labels, np.uint16)
    return synth.to_pcl(xyz, lab)


@pytest.mark.parametrize("case", ["one_class", "empty", "single_point", "labels_beyond_table", "mixed"])
def test_semantic_split_edges(oracle, gpu_ctx, case):
    """categoryMapping (semanticFusionNode.cpp:173-189) as one five-way partition: every class list equals the oracle's, in input order,
    when a class is empty, when all points fall into one class, and for labels the table does not hold (they land in `outlier`)."""
    import lisreg
    rng = np.random.default_rng(7)
    if case == "one_class":
        cloud = _labelled(5000, 18)                   # using_label[18] = 81: every point is a pole
    elif case == "empty":
        cloud = _labelled(0, 0)
    elif case == "single_point":
        cloud = _labelled(1, 9)
    elif case == "labels_beyond_table":
        cloud = _labelled(3000, rng.integers(20, 260, 3000))      # labels 20..259: & 31 picks a table entry, 20..31 have none
    else:
        cloud = _labelled(40000, rng.integers(0, 20, 40000))
    want = oracle.semantic_split(cloud)
    got = gpu_ctx.semantic_split(cloud)
    assert [len(g) for g in got] == [len(w) for w in want]
    for g, w in zip(got, want):
        for f in ("x", "y", "z", "intensity", "label"):               # field by field: the structs' padding bytes are nobody's
            assert np.array_equal(g[f], w[f]), f
    # the device-record form gives the same five clouds (label in the payload)
    n = len(cloud)
    if n:
        rec = lisreg.pack_device_records(cloud)
        src = lisreg.DeviceArray(rec)
        outs = [lisreg.DeviceArray(np.zeros((n, 4), np.float32)) for _ in range(5)]
        cnt = gpu_ctx.semantic_split_device(src.ptr, n, [o.ptr for o in outs], n)
        assert cnt == [len(w) for w in want]
        for o, c, w in zip(outs, cnt, want):
            assert np.array_equal(o.download(c)[:, :3], np.stack([w["x"], w["y"], w["z"]], 1).reshape(c, 3))


def test_voxel_multi_with_empty_and_tiny_clouds(oracle, gpu_ctx):
    """K clouds in one launch sequence: an empty cloud among them, a one-point cloud, a cloud whose points share one voxel — each output
    equals the single-cloud grid of the oracle bit for bit (the joint sort keeps every cloud's own geometry)."""
    import lisreg
    from lisreg import synth
    rng = np.random.default_rng(3)
    clouds = [_labelled(20000, rng.integers(0, 20, 20000), 1), _labelled(0, 0), _labelled(1, 5, 2),
              synth.to_pcl((rng.uniform(0, 0.05, (300, 3)) + 3.0).astype(np.float32), np.full(300, 4, np.uint16)),
              _labelled(7000, 13, 4)]
    leafs = [0.4, 0.2, 0.2, 0.5, 0.6]
    ins = [lisreg.DeviceArray(lisreg.pack_device_records(c) if len(c) else np.zeros((1, 4), np.float32)) for c in clouds]
    outs = [lisreg.DeviceArray(np.zeros((max(len(c), 1), 4), np.float32)) for c in clouds]
    nd = gpu_ctx.voxel_downsample_multi_device([a.ptr for a in ins], [len(c) for c in clouds], leafs, [o.ptr for o in outs],
                                               [max(len(c), 1) for c in clouds])
    for c, leaf, o, k in zip(clouds, leafs, outs, nd):
        if len(c) == 0:
            assert k == 0
            continue
        st, want = oracle.voxel_grid(c, leaf)
        assert st == 0 and k == len(want)
        got = o.download(k)
        assert np.array_equal(got[:, :3], np.stack([want["x"], want["y"], want["z"]], 1))
        assert np.array_equal(got[:, 3].view(np.uint32) & 0xffff, want["label"].astype(np.uint32))


def test_localmap_extract_with_empty_classes(oracle, gpu_ctx):
    """The fused crop + target assembly of lisreg_localmap_extract on a map whose dynamic and pole classes are empty (a first frame
    without cars or poles): counts, crop box and both targets equal the oracle chain's; then a box that cuts everything away."""
    import lisreg
    import replay_oracle as ro
    from lisreg import replay, synth
    frames = [c for c, _ in replay.synthetic_drive(2, h=32, w=900, car=False)]
    lm = lisreg.localmap_default_params()
    T0 = np.array([0, 0, 0.01, 0.2, -0.1, 0.0], np.float32)
    parts = gpu_ctx.semantic_split(frames[0])                  # dynamic, ground, building, pole, outlier
    keep = [parts[0][:0], parts[3][:0], parts[1], parts[2], parts[4][:0]]      # map order: dynamic, pole, ground, building, outlier
    gpu_ctx.localmap_reset(9)
    gpu_ctx.localmap_insert(9, keep, T0, lm)
    info = gpu_ctx.localmap_extract(9, T0, lm, target_slot=3)
    om = ro.LocalMapOracle(frames[0].dtype)
    om.insert(keep, T0)
    want_c, want_s, want_crop = om.extract(T0)
    assert info["n_target_corner"] == len(want_c) == 0
    assert info["n_target_surf"] == len(want_s) > 1000
    assert np.array_equal(info["crop"], want_crop)
    idx = gpu_ctx.target_index(3, 1)
    got = np.zeros((idx["n"], 3), np.float32); got[idx["sorted"][:, 3].view(np.int32)] = idx["sorted"][:, :3]
    assert np.array_equal(got, synth.pcl_xyz(want_s))
    far = np.array([0, 0, 0, 500.0, 500.0, 0.0], np.float32)        # the crop box 500 m away: nothing survives, nothing faults
    info = gpu_ctx.localmap_extract(9, far, lm, target_slot=3)
    assert info["n_target_surf"] == 0 and info["n_target_corner"] == 0 and sum(info["n"]) == 0


def test_concat_device_equals_numpy(gpu_ctx):
    """lisreg_concat_device: K clouds end to end (empty ones among them), stream-ordered before the next call of the context."""
    import lisreg
    rng = np.random.default_rng(5)
    parts = [rng.normal(size=(n, 4)).astype(np.float32) for n in (1000, 0, 1, 70001)]
    devs = [lisreg.DeviceArray(p if len(p) else np.zeros((1, 4), np.float32)) for p in parts]
    out = lisreg.DeviceArray(np.zeros((sum(len(p) for p in parts), 4), np.float32))
    tot = gpu_ctx.concat_device([d.ptr for d in devs], [len(p) for p in parts], out.ptr)
    assert tot == sum(len(p) for p in parts)
    gpu_ctx.upload_cloud(np.zeros(1, np.dtype({"names": ["x", "y", "z"], "formats": ["<f4"] * 3, "offsets": [0, 4, 8], "itemsize": 16})), devs[2].ptr)   # a later call: waits for the stream
    assert np.array_equal(out.download(tot).view(np.uint32), np.concatenate(parts).view(np.uint32))
    with pytest.raises(lisreg.LisregError):
        gpu_ctx.concat_device([0] * 9, [0] * 9, out.ptr)
