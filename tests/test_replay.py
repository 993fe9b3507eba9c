"""BASELINE configs[2] (sequential replay, semantic mask on) and configs[0] (scan-to-map odometry on raw, unlabelled sweeps) on
synthetic drives, and the device-resident sliding local map.

CPU (-m "not gpu"): the oracle chain (oracle/replay_oracle.py: the reference's frame loop restated over the C restatement's
primitives) tracks the known trajectory of a short drive; its pose-guess helper agrees with the library's host helper.
GPU: (1) lisreg_localmap_insert / _extract against the oracle's LocalMapOracle with IDENTICAL poses fed to both — every class
cloud, the bound, the crop box and both targets must then be bit-identical; (2) the HIP frame loop (lisreg.replay) against the
oracle frame loop, frame by frame: same iteration counts, poses within 1e-3 m / 1e-3 rad, and the trajectory file format."""
import os

import numpy as np
import pytest

from helpers import pose_err


def _records(cloud):
    import lisreg
    return lisreg.pack_device_records(cloud)


def test_oracle_chain_tracks_the_synthetic_drive(oracle):
    import replay_oracle as ro
    from lisreg import replay
    frames, truth = zip(*replay.synthetic_drive(5, h=16, w=450, step=0.15))
    recs = ro.replay(frames, n_threads=8)
    assert len(recs) == 5 and recs[0]["stats"] is None
    for rec, t in zip(recs[1:], truth[1:]):
        assert rec["stats"]["status"] == 0
        assert np.abs(rec["T"][3:5] - t[3:5]).max() < 0.05 and abs(rec["T"][2] - t[2]) < 0.01, (rec["frame"], rec["T"], t)
    # the local map is the union of what was inserted, cropped and down-sampled: non-empty static classes, no outliers
    assert recs[-1]["n_map"][1] > 0 and recs[-1]["n_map"][2] > 0 and recs[-1]["n_map"][3] > 0 and recs[-1]["n_map"][4] == 0
    assert recs[-1]["feature_point_num"] == sum(recs[-1]["n_map"])


def test_predict_pose_helper_matches_oracle():
    import lisreg
    import replay_oracle as ro
    rng = np.random.default_rng(3)
    for _ in range(20):
        a = np.concatenate([rng.uniform(-0.1, 0.1, 2), rng.uniform(-3, 3, 1), rng.uniform(-50, 50, 3)]).astype(np.float32)
        b = (a + np.concatenate([rng.uniform(-0.01, 0.01, 3), rng.uniform(-1, 1, 3)])).astype(np.float32)
        g1, g2 = lisreg.predict_pose(a, b), ro.predict_pose(a, b)
        assert np.array_equal(g1, g2), (g1, g2)                       # both are Eigen's float arithmetic step by step
    assert max(pose_err(lisreg.predict_pose(a, a), a)) < 1e-6          # no motion -> the same pose


@pytest.mark.parametrize("variant", [0, 1])
def test_update_initial_guess_follows_the_reference_branches(variant):
    """updateInitialGuess with odometry / IMU input (odomEstimationNode.cpp:297-419, subMapOptmizationNode.cpp:896-1032): random
    availability patterns through lisreg_update_initial_guess and through the branch-by-branch restatement — poses and
    transPredictionMapped equal to the bit after every call, including the first-odometry-message fall-through of copy #1."""
    import lisreg
    import replay_oracle as ro
    for seed in range(12):
        rng = np.random.default_rng(100 * variant + seed)
        heading = bool(seed & 1)
        g, o = lisreg.InitialGuess(variant, heading), ro.InitialGuessOracle(variant, heading)
        T = np.zeros(6, np.float32)
        imu = rng.uniform(-0.05, 0.05, 3); imu[2] = rng.uniform(-3, 3)
        odo = np.concatenate([rng.uniform(-20, 20, 3), rng.uniform(-0.05, 0.05, 2), rng.uniform(-3, 3, 1)])
        # patterns: all three kinds of drives + one that switches its inputs on and off
        mode = seed % 4
        for frame in range(25):
            imu = imu + rng.normal(0, 0.004, 3)
            odo = odo + np.concatenate([rng.normal(0.5, 0.1, 1), rng.normal(0, 0.05, 2), rng.normal(0, 0.003, 3)])
            oa = {0: False, 1: True, 2: False, 3: bool(rng.integers(0, 2))}[mode]
            ia = {0: False, 1: True, 2: True, 3: bool(rng.integers(0, 2))}[mode]
            if mode == 1 and frame < 3:
                oa = False                                           # odometry arrives a few frames late
            Tg, pg = g.update(T, oa, ia, imu, odo)
            To, po = o.update(T, oa, ia, imu, odo)
            assert np.array_equal(Tg, To), (variant, seed, frame, Tg, To)
            assert (pg is None) == (po is None) and (pg is None or np.array_equal(pg, po)), (variant, seed, frame)
            # what scan2SubMapOptimization would do next: move the pose a little (the registration's correction)
            T = (Tg + np.concatenate([rng.normal(0, 1e-3, 3), rng.normal(0, 0.02, 3)])).astype(np.float32)
    # the constant-velocity branch of the new entry point is lisreg_predict_pose
    g = lisreg.InitialGuess(variant)
    T0, _ = g.update(np.zeros(6, np.float32)); T1, _ = g.update(np.array([0, 0, 0.1, 1, 0, 0], np.float32))
    assert np.array_equal(T1, np.array([0, 0, 0.1, 1, 0, 0], np.float32))   # the first constant-velocity call only records
    T2, _ = g.update(np.array([0.01, 0, 0.2, 2, 0.5, 0], np.float32))
    assert np.array_equal(T2, lisreg.predict_pose(np.array([0, 0, 0.1, 1, 0, 0], np.float32), np.array([0.01, 0, 0.2, 2, 0.5, 0], np.float32)))


def test_trajectory_file_format(tmp_path):
    from lisreg import replay
    poses = [np.array([0, 0, 0.1 * k, 1.0 * k, 0.5 * k, 0], np.float32) for k in range(3)]
    p = tmp_path / "traj.txt"
    replay.write_trajectory(str(p), poses)
    rows = [l.split() for l in open(p).read().strip().split("\n")]
    assert len(rows) == 3 and all(len(r) == 12 for r in rows)
    first = np.array(rows[0], np.float64).reshape(3, 4)
    assert np.allclose(first, np.eye(4)[:3], atol=1e-9) and "e" in rows[1][0]          # H_init^-1 H, scientific notation


@pytest.mark.gpu
def test_localmap_composite_equals_oracle_bitwise(oracle, gpu_ctx):
    """insert / extract / insert ... with the SAME poses on both sides: the device-resident map must equal the host restatement
    exactly (class clouds incl. labels, feature_point_num, bound, crop box, targets) — including the dynamic removal, which
    starts once feature_point_num > 16000."""
    import lisreg
    import replay_oracle as ro
    from lisreg import replay
    frames, truth = zip(*replay.synthetic_drive(6, h=32, w=900))
    lm = None
    P = lisreg.localmap_default_params()
    gpu_ctx.localmap_reset(3)
    removed_any = False
    for k, (cloud, T) in enumerate(zip(frames, truth)):
        T = T.astype(np.float32)
        parts = oracle.semantic_split(cloud)
        full = dict(dynamic=parts[0], ground=parts[1], building=parts[2], pole=parts[3], outlier=parts[4])
        clouds = [full[c] for c in ro.CLASSES]
        if lm is None:
            lm = ro.LocalMapOracle(cloud.dtype)
        if k > 0:
            tc, ts, isect = lm.extract(T)
            info = gpu_ctx.localmap_extract(3, T, P, target_slot=0)
            assert np.array_equal(info["crop"], isect)
            assert info["n_target_corner"] == len(tc) and info["n_target_surf"] == len(ts)
            assert np.array_equal(gpu_ctx.localmap_get(3, 5), _records(tc))
            assert np.array_equal(gpu_ctx.localmap_get(3, 6), _records(ts))
        n_dyn_before = len(lm.cls[0])
        lm.insert(clouds, T)
        removed_any |= len(lm.cls[0]) - n_dyn_before < len(full["dynamic"])
        info = gpu_ctx.localmap_insert(3, clouds, T, P)
        assert info["n"] == [len(c) for c in lm.cls] and info["feature_point_num"] == lm.feature_point_num
        assert np.array_equal(info["bound"], lm.bound)
        for c in range(5):
            assert np.array_equal(gpu_ctx.localmap_get(3, c), _records(lm.cls[c])), (k, c)
    assert removed_any, "the drive never exercised the map-based dynamic removal"
    # the extracted target is a registration target like any other
    T, st, _ = gpu_ctx.align(*[lisreg.synth.to_pcl(np.zeros((0, 3), np.float32), np.zeros(0, np.uint16))] * 2, truth[-1].astype(np.float32),
                             lisreg.default_params(2))
    assert st["status"] == lisreg.NOT_ENOUGH_FEATURES


@pytest.mark.gpu
def test_sequential_replay_matches_oracle_frame_by_frame(oracle):
    """configs[2] on the synthetic drive at full scan size: 20 frames of 64x1800 labelled sweeps, label-weighted copy #2
    registration against the sliding local map, early exit — HIP chain vs oracle chain."""
    import lisreg
    import replay_oracle as ro
    from lisreg import replay
    n = 20
    frames, truth = zip(*replay.synthetic_drive(n))
    ref = ro.replay(frames, n_threads=16)
    ctx = lisreg.Context(0)
    got = replay.replay(ctx, frames)
    ctx.close()
    assert len(got) == len(ref) == n
    worst = 0.0
    for g, r, t in zip(got, ref, truth):
        if r["stats"] is None:
            assert g["stats"] is None
            continue
        assert g["stats"]["status"] == r["stats"]["status"] == 0, (g["frame"], g["stats"], r["stats"])
        assert abs(g["stats"]["iters"] - r["stats"]["iters"]) <= 1, (g["frame"], g["stats"], r["stats"])
        e = max(pose_err(g["T"], r["T"]))
        worst = max(worst, e)
        assert e <= 1e-3, (g["frame"], e, g["T"], r["T"])
        assert max(pose_err(g["guess"], r["guess"])) <= 1e-3
        # map sizes: identical up to the points a sub-0.1-mm pose difference moves across a voxel / crop boundary
        for a, b in zip(g["n_map"], r["n_map"]):
            assert abs(a - b) <= max(3, 0.002 * b), (g["frame"], g["n_map"], r["n_map"])
        assert abs(g["n_target_surf"] - r["n_target_surf"]) <= max(3, 0.002 * r["n_target_surf"])
        assert np.abs(np.asarray(g["T"], np.float64)[3:5] - t[3:5]).max() < 0.05          # and both follow the drive
    print(f"replay: worst pose difference HIP vs oracle over {n} frames: {worst:.2e}; iterations per frame "
          f"{[g['stats']['iters'] for g in got[1:]]}")


def _write_kitti_dir(root, frames):
    """frames as a KITTI odometry / SemanticKITTI tree: velodyne/%06d.bin (x y z remission), labels/%06d.label (uint32)."""
    inv = {1: 10, 9: 40, 13: 50, 18: 80}                 # RangeNet class -> a SemanticKITTI raw id that learning_map sends back to it
    vd, ld = os.path.join(root, "sequences", "05", "velodyne"), os.path.join(root, "sequences", "05", "labels")
    os.makedirs(vd); os.makedirs(ld)
    for k, cloud in enumerate(frames):
        raw = np.zeros((len(cloud), 4), np.float32)
        raw[:, 0], raw[:, 1], raw[:, 2] = cloud["x"], cloud["y"], cloud["z"]
        raw.tofile(os.path.join(vd, f"{k:06d}.bin"))
        sem = np.array([inv[int(l)] for l in cloud["label"]], np.uint32) | (np.uint32(7) << 16)       # instance id in the high half
        sem.tofile(os.path.join(ld, f"{k:06d}.label"))


def test_kitti_reader_round_trip(tmp_path):
    from lisreg import replay
    frames, _ = zip(*replay.synthetic_drive(2, h=16, w=450))
    _write_kitti_dir(str(tmp_path), frames)
    got = [c for c, _ in replay.kitti_sequence(str(tmp_path), "05")]
    assert len(got) == 2
    for a, b in zip(got, frames):
        assert a.dtype == b.dtype and np.array_equal(a["label"], b["label"])
        assert np.array_equal(a["x"], b["x"]) and np.array_equal(a["z"], b["z"])
    import subprocess, sys
    tool = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "kitti_replay.py")
    rc = subprocess.call([sys.executable, tool, "--root", str(tmp_path / "nowhere"), "--seq", "05"], stderr=subprocess.DEVNULL)
    assert rc == 2                                         # no dataset -> says so, no traceback, no fake numbers


@pytest.mark.gpu
def test_kitti_tool_end_to_end_on_a_synthesised_sequence(tmp_path):
    """tools/kitti_replay.py on a KITTI-format directory written from the synthetic drive: it must reproduce lisreg.replay on
    the same frames bit for bit, write the trajectory in the reference's format and agree with the CPU restatement."""
    import json
    import subprocess
    import sys
    import lisreg
    from lisreg import replay
    frames, truth = zip(*replay.synthetic_drive(6, h=32, w=900))
    _write_kitti_dir(str(tmp_path), frames)
    tool = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "kitti_replay.py")
    out = tmp_path / "traj.txt"
    res = subprocess.run([sys.executable, tool, "--root", str(tmp_path), "--seq", "05", "--out", str(out), "--check-oracle", "6"],
                         capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
    summary = json.loads(res.stdout.strip().split("\n")[-1])
    assert summary["frames"] == 6 and summary["oracle_check"]["max_trans_diff_m"] <= 1e-3 and summary["oracle_check"]["max_rot_diff_rad"] <= 1e-3
    ctx = lisreg.Context(0)
    direct = replay.replay(ctx, frames)
    ctx.close()
    assert np.allclose(summary["final_pose"], direct[-1]["T"], rtol=0, atol=0)
    # --exact: the exact-arithmetic build in the loop -> every checked frame equal to the restatement's to the bit
    res = subprocess.run([sys.executable, tool, "--root", str(tmp_path), "--seq", "05", "--check-oracle", "6", "--exact"], capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
    chk = json.loads(res.stdout.strip().split("\n")[-1])["oracle_check"]
    assert chk["frames_bit_identical"] == chk["frames"] == 6 and chk["max_trans_diff_m"] == 0.0
    rows = np.loadtxt(out)
    assert rows.shape == (6, 12) and np.allclose(rows[0].reshape(3, 4), np.eye(4)[:3])
    assert abs(rows[-1, 3] - truth[-1][3]) < 0.05 and abs(rows[-1, 7] - truth[-1][4]) < 0.05


# ---- the odometry loop on raw sweeps (odomEstimationNode, copy #1; BASELINE configs[0]) -------------------------------------
def test_kitti_ring_assignment():
    """laserPretreatmentNode.cpp:95-117: the two-slope HDL-64 table, the angle / ring limits, the range filter, NaN removal."""
    from lisreg import replay
    ang = np.array([1.9, 0.0, -8.0, -8.83, -9.0, -20.0, -24.3, -24.4, 2.1, -3.0], np.float64)
    r = 10.0
    raw = np.zeros((len(ang) + 3, 4), np.float32)
    raw[:len(ang), 0] = r * np.cos(np.radians(ang)); raw[:len(ang), 2] = r * np.sin(np.radians(ang))
    raw[len(ang)] = (80.0, 0, 0, 0)                       # beyond lidarMaxRange
    raw[len(ang) + 1] = (np.nan, 0, 0, 0)
    raw[len(ang) + 2] = (5.0, 5.0, -1.0, 0.5)
    out = replay.kitti_rings(raw)
    a32 = np.degrees(np.arctan(raw[:len(ang), 2] / raw[:len(ang), 0])).astype(np.float32)
    want = [int((2 - a) * 3.0 + 0.5) if a >= np.float32(-8.83) else 32 + int((np.float32(-8.83) - a) * 2.0 + 0.5) for a in a32]
    keep = [(a <= 2) and (a >= np.float32(-24.33)) and 0 <= w <= 50 for a, w in zip(a32, want)]
    assert list(out["ring"][: sum(keep)]) == [w for w, k in zip(want, keep) if k]
    assert len(out) == sum(keep) + 1 and out["intensity"][-1] == np.float32(0.5)        # the far and the NaN point are gone
    assert not keep[7] and not keep[8]                                                   # -24.4 deg and 2.1 deg are outside


def test_oracle_odometry_chain_tracks_the_raw_drive(oracle):
    import replay_oracle as ro
    from lisreg import replay
    h, w = 32, 900
    frames, truth = zip(*replay.synthetic_raw_drive(8, h=h, w=w))
    recs = ro.replay_odom(frames, oracle.FeatureParams(h, w, 1, 0.0, 70.0, 1.0, 0.1), n_threads=8)
    assert recs[0]["keyframe"] and recs[0]["stats"] is None
    for rec, t in zip(recs[1:], truth[1:]):
        assert rec["stats"]["status"] == 0 and rec["stats"]["degenerate"] == 0
        assert np.abs(rec["T"][3:5] - t[3:5]).max() < 0.06 and abs(rec["T"][2] - t[2]) < 0.01, (rec["frame"], rec["T"], t)
    assert not all(r["keyframe"] for r in recs)             # the keyframe gate closed at least once (key_id > 5, < 1.4 m travelled)
    assert recs[-1]["n_target_surf"] > recs[1]["n_target_surf"]        # the target is the union of the keyframes


@pytest.mark.gpu
def test_odometry_replay_matches_oracle_frame_by_frame(oracle):
    """configs[0] stand-in at full sweep size: 64x1800 raw sweeps -> projection + features -> voxel grids -> copy #1 registration
    against the <= 19 newest keyframes -> keyframe gate; HIP chain vs oracle chain."""
    import lisreg
    import replay_oracle as ro
    from lisreg import replay
    n = 9
    frames, truth = zip(*replay.synthetic_raw_drive(n))
    ref = ro.replay_odom(frames, n_threads=16)
    ctx = lisreg.Context(0)
    got = replay.replay_odom(ctx, frames)
    ctx.close()
    worst = 0.0
    for g, r, t in zip(got, ref, truth):
        assert (g["n_corner"], g["n_surf"]) == (r["n_corner"], r["n_surf"])             # feature extraction is exact
        assert g["keyframe"] == r["keyframe"] and g["key_id"] == r["key_id"]
        if r["stats"] is None:
            assert g["stats"] is None
            continue
        assert g["stats"]["status"] == r["stats"]["status"] == 0
        assert abs(g["stats"]["iters"] - r["stats"]["iters"]) <= 1, (g["frame"], g["stats"], r["stats"])
        e = max(pose_err(g["T"], r["T"]))
        worst = max(worst, e)
        assert e <= 1e-3, (g["frame"], e, g["T"], r["T"])
        assert abs(g["n_target_surf"] - r["n_target_surf"]) <= max(3, 0.002 * r["n_target_surf"])
        assert g["n_src_surf"] == r["n_src_surf"] and g["n_src_corner"] == r["n_src_corner"]   # same input, same voxel grid
        assert np.abs(np.asarray(g["T"], np.float64)[3:5] - t[3:5]).max() < 0.06
    print(f"odometry replay: worst pose difference HIP vs oracle over {n} frames: {worst:.2e}; iterations "
          f"{[g['stats']['iters'] for g in got[1:]]}; keyframes {[int(g['keyframe']) for g in got]}")


@pytest.mark.gpu
def test_kitti_tool_odometry_mode_on_a_synthesised_sequence(tmp_path):
    """tools/kitti_replay.py --mode odom on velodyne/*.bin files only (no labels): ring assignment -> features -> keyframe
    odometry; must agree with the CPU restatement of the same loop and write the trajectory in the reference's format."""
    import json
    import subprocess
    import sys
    from lisreg import replay
    frames, truth = zip(*replay.synthetic_raw_drive(6))
    vd = tmp_path / "sequences" / "00" / "velodyne"
    os.makedirs(vd)
    for k, sw in enumerate(frames):
        raw = np.stack([sw["x"], sw["y"], sw["z"], sw["intensity"]], 1).astype(np.float32)
        raw.tofile(str(vd / f"{k:06d}.bin"))
    tool = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "kitti_replay.py")
    out = tmp_path / "traj.txt"
    res = subprocess.run([sys.executable, tool, "--root", str(tmp_path), "--seq", "00", "--mode", "odom", "--out", str(out),
                          "--check-oracle", "6"], capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
    summary = json.loads(res.stdout.strip().split("\n")[-1])
    assert summary["frames"] == 6 and summary["mode"] == "odom"
    assert summary["oracle_check"]["max_trans_diff_m"] <= 1e-3 and summary["oracle_check"]["max_rot_diff_rad"] <= 1e-3
    rows = np.loadtxt(out)
    assert rows.shape == (6, 12) and np.allclose(rows[0].reshape(3, 4), np.eye(4)[:3])
    assert abs(rows[-1, 3] - truth[-1][3]) < 0.08 and abs(rows[-1, 7] - truth[-1][4]) < 0.08


@pytest.mark.gpu
def test_keyframe_ring_target_equals_oracle_bitwise(oracle, gpu_ctx):
    """lisreg_keyframes_push / _target (the odometry node's target, kept in HBM) against the host restatement of
    saveKeyFrames + laserCloudInfoHandler: transformPointCloud, newest-first concatenation, the two voxel grids; and the ring
    drops its oldest frames beyond max_keep."""
    import replay_oracle as ro
    from lisreg import replay, synth
    frames, truth = zip(*replay.synthetic_raw_drive(5, h=32, w=900))
    fp = oracle.FeatureParams(32, 900, 1, 0.0, 70.0, 1.0, 0.1)
    gpu_ctx.keyframes_reset(2)
    kc, ks = [], []
    for k, (sw, T) in enumerate(zip(frames, truth)):
        f = oracle.extract_features(sw, fp)
        corner, surf = ro._xyzi(sw, f["corner"]), ro._xyzi(sw, f["surface"])
        T = T.astype(np.float32)
        info = gpu_ctx.keyframes_push(2, corner, surf, T, max_keep=3)
        kc.append(oracle.transform_cloud(corner, T)); ks.append(oracle.transform_cloud(surf, T))
        kc, ks = kc[-3:], ks[-3:]
        assert info["n_keyframes"] == len(kc)
        tinfo = gpu_ctx.keyframes_target(2, 0.2, 0.4, target_slot=0)
        want_c = oracle.voxel_grid(ro.cat(kc[::-1]), 0.2)[1]
        want_s = oracle.voxel_grid(ro.cat(ks[::-1]), 0.4)[1]
        assert (tinfo["n_target_corner"], tinfo["n_target_surf"]) == (len(want_c), len(want_s))
        for kind, want in ((0, want_c), (1, want_s)):
            idx = gpu_ctx.target_index(0, kind)
            got = np.zeros((idx["n"], 3), np.float32)
            got[idx["sorted"][:, 3].view(np.int32)] = idx["sorted"][:, :3]          # back to the target cloud's own order
            assert np.array_equal(got, synth.pcl_xyz(want)), (k, kind)
    # the built target is reused while the ring is unchanged (no key frame since the last call, same leaves, the slot still holds it)
    # and rebuilt as soon as one of the three changes: another call, another cloud in the slot in between, other leaf sizes
    def slot0():
        return [gpu_ctx.target_index(0, kind) for kind in (0, 1)]
    first = slot0()
    again = gpu_ctx.keyframes_target(2, 0.2, 0.4, target_slot=0)
    assert again == tinfo and all(np.array_equal(a["sorted"], b["sorted"]) for a, b in zip(first, slot0()))
    other_c, other_s = synth.make_submap(3000, 5)
    gpu_ctx.set_target(other_c, other_s)                                            # slot 0 now holds something else
    assert gpu_ctx.target_index(0, 1)["n"] == len(other_s)
    assert gpu_ctx.keyframes_target(2, 0.2, 0.4, target_slot=0) == tinfo           # ... and is rebuilt from the ring
    assert all(np.array_equal(a["sorted"], b["sorted"]) for a, b in zip(first, slot0()))
    coarse = gpu_ctx.keyframes_target(2, 0.4, 0.8, target_slot=0)
    assert coarse["n_target_surf"] == len(oracle.voxel_grid(ro.cat(ks[::-1]), 0.8)[1]) < tinfo["n_target_surf"]


@pytest.mark.gpu
def test_device_resident_frame_loop_equals_host_driven_loop_bitwise():
    """DeviceReplayer (sweep uploaded once, every cloud of the frame stays in HBM) against Replayer (host clouds between the
    steps): the same kernels in the same order, so poses, iteration counts and map sizes are identical."""
    import lisreg
    from lisreg import replay
    frames, _ = zip(*replay.synthetic_drive(8, h=32, w=900))
    ctx = lisreg.Context(0)
    a = replay.replay(ctx, frames)
    b = replay.replay(ctx, frames, device_resident=True)
    ctx.close()
    for x, y in zip(a, b):
        assert np.array_equal(x["T"], y["T"]) and x["n_map"] == y["n_map"]
        assert (x["stats"] is None) == (y["stats"] is None)
        if x["stats"]:
            assert x["stats"]["iters"] == y["stats"]["iters"] and x["n_src_surf"] == y["n_src_surf"]


@pytest.mark.gpu
def test_device_resident_odometry_loop_equals_host_driven_loop_bitwise():
    import lisreg
    from lisreg import replay
    frames, _ = zip(*replay.synthetic_raw_drive(8))
    ctx = lisreg.Context(0)
    a = replay.replay_odom(ctx, frames)
    b = replay.replay_odom(ctx, frames, device_resident=True)
    ctx.close()
    for x, y in zip(a, b):
        assert np.array_equal(x["T"], y["T"]) and x["keyframe"] == y["keyframe"] and (x["n_corner"], x["n_surf"]) == (y["n_corner"], y["n_surf"])
        if x["stats"]:
            assert x["stats"]["iters"] == y["stats"]["iters"] and x["n_src_surf"] == y["n_src_surf"] and x["n_target_surf"] == y["n_target_surf"]


# ---- copy #3: submap_t + insert_submap + extractSubMapCloud + subMap2SubMapOptimization ---------------------------------------------
def _submap_drive(n_submaps=3, per=3, h=32, w=900):
    """Key frames of a synthetic labelled drive grouped into submaps of `per` frames: per frame the five DOWN-sampled class clouds
    (keyframeInit's grids) and its true pose; a submap's pose = its first frame's."""
    import replay_oracle as ro
    from lisreg import replay
    frames, truth = zip(*replay.synthetic_drive(n_submaps * per, h=h, w=w, step=0.9))
    out = []
    for s in range(n_submaps):
        ks = list(range(s * per, (s + 1) * per))
        downs = []
        for k in ks:
            f = frames[k].copy()
            g = np.flatnonzero(f["label"] == 9)
            f["label"][g[::7]] = 15                   # some "vegetation": the outlier class (label.yaml:177-196), which insert_submap keeps
            _, down = ro.split_and_downsample(f)
            downs.append([down[c] for c in ro.CLASSES])
        out.append(dict(frames=downs, poses=[truth[k].astype(np.float32) for k in ks]))
    return out


def _relative(T_sub, T_frame):
    """relative_pose of a key frame inside its submap: T_sub^-1 * T_frame (host float64 here; both sides are fed the same numbers)."""
    from lisreg import synth
    M = np.linalg.inv(synth.pose_matrix(T_sub)) @ synth.pose_matrix(T_frame)
    return np.array([np.arctan2(M[2, 1], M[2, 2]), np.arcsin(-M[2, 0]), np.arctan2(M[1, 0], M[0, 0]), M[0, 3], M[1, 3], M[2, 3]], np.float32)


def test_submap_crop_boxes_helper_matches_oracle():
    import ctypes as C
    import lisreg
    import replay_oracle as ro
    rng = np.random.default_rng(11)
    L = lisreg.lib()
    for _ in range(20):
        pb = np.sort(rng.uniform(-40, 40, (2, 3)), 0).ravel(); cb = np.sort(rng.uniform(-40, 40, (2, 3)), 0).ravel()
        Tp = np.concatenate([rng.uniform(-0.05, 0.05, 2), rng.uniform(-3, 3, 1), rng.uniform(-30, 30, 3)]).astype(np.float32)
        Tc = (Tp + np.concatenate([rng.uniform(-0.02, 0.02, 3), rng.uniform(-5, 5, 3)])).astype(np.float32)
        a, b = np.zeros(6), np.zeros(6)
        dp = C.POINTER(C.c_double); fp = C.POINTER(C.c_float)
        L.lisreg_submap_crop_boxes(pb.ctypes.data_as(dp), Tp.ctypes.data_as(fp), cb.ctypes.data_as(dp), Tc.ctypes.data_as(fp), 10.0,
                                   a.ctypes.data_as(dp), b.ctypes.data_as(dp))
        ia, ib = ro.submap_crop_boxes(pb, Tp, cb, Tc, 10.0)
        assert np.array_equal(a, ia) and np.array_equal(b, ib), (a - ia, b - ib)


@pytest.mark.gpu
def test_submap_composite_equals_oracle_bitwise(oracle, gpu_ctx):
    """fisrt_submap / insert_submap for three submaps and extractSubMapCloud for the two consecutive pairs, the SAME poses on both
    sides: class clouds (all five, outlier included), feature_point_num, local and global bound, both crop boxes, both targets and
    both down-sampled sources bit for bit."""
    import lisreg
    import replay_oracle as ro
    subs = _submap_drive(3, 3)
    P = lisreg.localmap_default_params()
    P.max_num_pts = 20000                      # so that the map-based dynamic removal (feature_point_num > 4000) is exercised
    oracles, removed = [], False
    for s, sub in enumerate(subs):
        mid = 20 + s
        gpu_ctx.localmap_reset(mid)
        so = ro.SubMapOracle(sub["frames"][0][0].dtype)
        T_sub = sub["poses"][0]
        for j, (clouds, T) in enumerate(zip(sub["frames"], sub["poses"])):
            rel = None if j == 0 else _relative(T_sub, T)
            n_dyn = len(so.cls[0])
            so.insert(clouds, rel, max_num_pts=20000)
            removed |= j > 0 and len(so.cls[0]) - n_dyn < len(clouds[0])
            info = gpu_ctx.submap_insert(mid, clouds, rel, T_sub, P)
            assert info["n"] == [len(c) for c in so.cls] and info["feature_point_num"] == so.feature_point_num
            assert info["n"][4] > 0                                                    # the outlier class IS kept here
            assert np.array_equal(info["local_bound"], so.bound) and np.array_equal(info["bound"], so.global_bound(T_sub))
            for c in range(5):
                assert np.array_equal(gpu_ctx.localmap_get(mid, c), _records(so.cls[c])), (s, j, c)
        oracles.append((so, T_sub))
    assert removed, "the drive never exercised the map-based dynamic removal"
    for s in (1, 2):
        (pre, Tp), (cur, Tc) = oracles[s - 1], oracles[s]
        guess = (Tc + np.array([0.004, -0.003, 0.01, 0.15, -0.1, 0.02], np.float32)).astype(np.float32)
        tc, ts, sc, ss, isect, isect_local = ro.extract_submap_cloud(pre, cur, Tp, guess)
        out = gpu_ctx.submap_extract(20 + s - 1, 20 + s, Tp, guess, target_slot=1)
        assert np.array_equal(out["isect"], isect) and np.array_equal(out["isect_local"], isect_local)
        assert (out["n_target_corner"], out["n_target_surf"], out["n_src_corner"], out["n_src_surf"]) == (len(tc), len(ts), len(sc), len(ss))
        assert len(ts) > 1000 and len(ss) > 500
        assert np.array_equal(gpu_ctx.localmap_get(20 + s - 1, 5), _records(tc)) and np.array_equal(gpu_ctx.localmap_get(20 + s - 1, 6), _records(ts))
        assert np.array_equal(gpu_ctx.localmap_get(20 + s, 5), _records(sc)) and np.array_equal(gpu_ctx.localmap_get(20 + s, 6), _records(ss))


@pytest.mark.gpu
@pytest.mark.parametrize("exact", [False, True])
def test_submap_to_submap_chain_matches_oracle(oracle, exact):
    """Variant 3 as a sequence (subMapOptmizationThread): for every new submap extractSubMapCloud against the previous one, then
    subMap2SubMapOptimization (copy #3: label weights, 30 iterations at most, convergence 0.002 deg / 0.02 cm, corner stage skipped when
    the target has no poles) from a perturbed guess — HIP chain (device-resident, sources handed over as device records) vs oracle
    chain; the registered pose becomes the submap's pose for the next pair on both sides.  With the exact-arithmetic build the chain is
    the oracle's to the bit: poses, iteration counts, target and source sizes."""
    import lisreg
    import replay_oracle as ro
    from helpers import copy_params
    subs = _submap_drive(4, 3, h=64, w=900)
    P = lisreg.localmap_default_params()
    p_o = oracle.default_params(3)
    p_g = copy_params(p_o, lisreg.Params)
    ctx = lisreg.Context(0)
    ctx.set_option("exact_arithmetic", 1 if exact else 0)
    pose_o, pose_g, worst = None, None, 0.0
    pre_o = None
    for s, sub in enumerate(subs):
        ctx.localmap_reset(s)
        so = ro.SubMapOracle(sub["frames"][0][0].dtype)
        T_true = sub["poses"][0]
        for j, (clouds, T) in enumerate(zip(sub["frames"], sub["poses"])):
            rel = None if j == 0 else _relative(T_true, T)
            so.insert(clouds, rel)
            ctx.submap_insert(s, clouds, rel, T_true, P)
        if s == 0:
            pose_o = pose_g = T_true.copy()
        else:
            guess = (T_true + np.array([0.003, -0.002, 0.008, 0.12, -0.08, 0.015], np.float32)).astype(np.float32)
            tc, ts, sc, ss, _, _ = ro.extract_submap_cloud(pre_o, so, pose_o, guess)
            To, st_o, _ = oracle.align(tc, ts, sc, ss, guess, p_o, n_threads=8, max_trace=1)
            out = ctx.submap_extract(s - 1, s, pose_g, guess, target_slot=0)
            Tg, st_g = ctx.align_device(out["src_corner_ptr"], out["n_src_corner"], out["src_surf_ptr"], out["n_src_surf"], guess, p_g)
            assert st_g["status"] == st_o["status"] == 0, (s, st_g, st_o)
            assert abs(st_g["iters"] - st_o["iters"]) <= 2, (s, st_g, st_o)
            assert abs(out["n_target_surf"] - len(ts)) <= max(3, 0.002 * len(ts)) and abs(out["n_src_surf"] - len(ss)) <= max(3, 0.002 * len(ss))
            e = max(pose_err(Tg, To))
            worst = max(worst, e)
            if exact:
                assert np.array_equal(np.asarray(Tg, np.float32), np.asarray(To, np.float32)) and st_g["iters"] == st_o["iters"], (s, Tg, To)
                assert (out["n_target_corner"], out["n_target_surf"], out["n_src_corner"], out["n_src_surf"]) == (len(tc), len(ts), len(sc), len(ss))
            assert e <= 1e-3, (s, e, Tg, To)
            assert np.abs(np.asarray(Tg, np.float64)[3:5] - T_true[3:5]).max() < 0.05 and abs(float(Tg[2]) - float(T_true[2])) < 0.01
            pose_o, pose_g = To.astype(np.float32), Tg.astype(np.float32)
        pre_o = so
    ctx.close()
    print(f"submap chain: worst pose difference HIP vs oracle over {len(subs) - 1} submap pairs: {worst:.2e}")


@pytest.mark.gpu
@pytest.mark.parametrize("device_resident", [True, False])
def test_exact_frame_loop_equals_oracle_chain_bitwise(oracle, device_resident):
    """The whole frame loop of configs[2] with the registration in the exact-arithmetic build — semantic split, per-class voxel grids,
    constant-velocity guess, sliding local map (voxel grids, crop, targets), label-weighted registration, insert — against the oracle
    chain: the pose of EVERY frame equal to the bit, iteration counts and correspondence counts equal, class sizes of the map equal."""
    import lisreg
    import replay_oracle as ro
    from lisreg import replay
    frames = [c for c, _ in replay.synthetic_drive(20, h=32, w=900)]
    ref = ro.replay(frames, n_threads=16)
    ctx = lisreg.Context(0)
    ctx.set_option("exact_arithmetic", 1)
    got = replay.replay(ctx, frames, device_resident=device_resident)
    ctx.close()
    for a, b in zip(got, ref):
        assert np.array_equal(np.asarray(a["T"], np.float32), np.asarray(b["T"], np.float32)), a["frame"]
        if a["stats"]:
            assert a["stats"]["iters"] == b["stats"]["iters"] and a["stats"]["n_corr_last"] == b["stats"]["n_corr_last"], a["frame"]
            assert (a["n_target_corner"], a["n_target_surf"], a["n_src_corner"], a["n_src_surf"]) == \
                   (b["n_target_corner"], b["n_target_surf"], b["n_src_corner"], b["n_src_surf"]), a["frame"]


@pytest.mark.gpu
@pytest.mark.parametrize("pattern", ["odom+imu", "imu", "mixed"])
def test_exact_frame_loop_with_imu_and_odometry_guess_equals_oracle_chain_bitwise(oracle, pattern):
    """The frame loop with cloudInfo.odomAvailable / imuAvailable set: updateInitialGuess's odometry-increment and IMU-increment branches
    (subMapOptmizationNode.cpp:928-982) feed the registration, and the IMU attitude enters transformUpdate (:1980-2001) — HIP chain in the
    exact build vs the oracle chain, every pose equal to the bit."""
    import lisreg
    import replay_oracle as ro
    from lisreg import replay, synth
    n = 12
    frames, truth = zip(*replay.synthetic_drive(n, h=32, w=900))
    rng = np.random.default_rng(77)
    gi = []
    for k in range(n):
        t = np.asarray(truth[k], np.float64)                                  # {roll, pitch, yaw, x, y, z} of the drive
        odo = (float(t[3] + 5.0), float(t[4] - 2.0), float(t[5] + 0.3), float(t[0] + rng.normal(0, 2e-4)), float(t[1] + rng.normal(0, 2e-4)),
               float(t[2] + rng.normal(0, 2e-4)))                              # the pre-integration guess lives in ITS OWN (shifted) frame: only increments count
        imu = (float(t[0] + rng.normal(0, 3e-4)), float(t[1] + rng.normal(0, 3e-4)), float(t[2] + rng.normal(0, 3e-4)))
        oa = {"odom+imu": k >= 2, "imu": False, "mixed": k % 3 != 1}[pattern]
        ia = {"odom+imu": True, "imu": True, "mixed": k % 4 != 2}[pattern]
        gi.append(dict(odom_available=oa, imu_available=ia, imu_rpy=imu, initial_guess=odo))
    ref = ro.replay(frames, n_threads=16, guess_inputs=gi)
    ctx = lisreg.Context(0)
    ctx.set_option("exact_arithmetic", 1)
    got = replay.replay(ctx, frames, guess_inputs=gi)
    ctx.close()
    for a, b, t in zip(got, ref, truth):
        assert np.array_equal(np.asarray(a["guess"], np.float32), np.asarray(b["guess"], np.float32)), a["frame"]
        assert np.array_equal(np.asarray(a["T"], np.float32), np.asarray(b["T"], np.float32)), a["frame"]
        if a["stats"]:
            assert a["stats"]["iters"] == b["stats"]["iters"] and a["stats"]["n_corr_last"] == b["stats"]["n_corr_last"], a["frame"]
            assert np.abs(np.asarray(a["T"], np.float64)[3:5] - np.asarray(t)[3:5]).max() < 0.1, (a["frame"], a["T"], t)     # and it follows the drive


@pytest.mark.gpu
def test_exact_odometry_loop_equals_oracle_chain_bitwise(oracle):
    """The raw-sweep odometry loop of configs[0] (range image + LOAM features, key-frame ring target, voxel grids, copy #1 in the
    exact-arithmetic build, key-frame gate) against the oracle chain: every pose equal to the bit, the same key-frame decisions."""
    import lisreg
    import replay_oracle as ro
    from lisreg import replay
    sweeps = [c for c, _ in replay.synthetic_raw_drive(10, h=32, w=900)]
    fp_o = oracle.FeatureParams(32, 900, 1, 0.0, 70.0, 1.0, 0.1)
    fp_g = lisreg.FeatureParams(32, 900, 1, 0.0, 70.0, 1.0, 0.1)
    ref = ro.replay_odom(sweeps, fp_o, n_threads=16)
    ctx = lisreg.Context(0)
    ctx.set_option("exact_arithmetic", 1)
    got = replay.replay_odom(ctx, sweeps, fp_g, device_resident=True)
    ctx.close()
    for a, b in zip(got, ref):
        assert np.array_equal(np.asarray(a["T"], np.float32), np.asarray(b["T"], np.float32)), a["frame"]
        assert a["keyframe"] == b["keyframe"] and (a["n_corner"], a["n_surf"]) == (b["n_corner"], b["n_surf"]), a["frame"]
