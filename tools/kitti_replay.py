#!/usr/bin/env python3
"""Sequential replay of a KITTI odometry / SemanticKITTI sequence on the HIP path (BASELINE.json configs[0] / configs[2]).

    python tools/kitti_replay.py --root $KITTI_ROOT --seq 05 [--frames 200] [--out traj_05.txt] [--check-oracle 5]
    python tools/kitti_replay.py --root $KITTI_ROOT --seq 00 --mode odom ...        (no labels needed)

Expects the standard layout <root>/sequences/<seq>/velodyne/*.bin (float32 x y z remission) and, for the semantic mask,
<root>/sequences/<seq>/labels/*.label (SemanticKITTI; mapped to the 20 RangeNet++ classes through the reference's
config/label.yaml learning_map, which is what the reference's semantic node publishes).  Every frame runs the loop of
SubMapOptmizationNode::makeSubMapThread (/root/reference/src/node/subMapOptmizationNode.cpp:597-755) through liblisreg
(lisreg.replay.Replayer): semantic split -> per-class voxel grids -> constant-velocity guess -> device-resident sliding local
map -> label-weighted registration (copy #2) with early exit -> map insert.  The trajectory is written in the reference's
format (transformFusion, :5079-5179: 12 numbers per pose, relative to the first, scientific notation) so that it can be scored
with the KITTI devkit exactly like the reference's result file.  --check-oracle N additionally runs the CPU restatement of
the same loop on the first N frames and reports the pose differences (the parity bar is 1e-3 m / 1e-3 rad).

--mode odom runs the scan-to-map odometry of odomEstimationNode instead (/root/reference/src/node/odomEstimationNode.cpp:164-232,
BASELINE configs[0]): ring assignment of laserPretreatmentNode -> range image + LOAM features -> voxel grids -> registration
(copy #1) against the <= 19 newest keyframes -> keyframe gate (lisreg.replay.OdomReplayer); it needs velodyne/*.bin only.

No dataset ships with this repository and none is reachable from the build environment; tests/test_replay.py exercises this
tool on a KITTI-format directory it synthesises.  Exit code 2 when the sequence directory does not exist."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lis-slam_amd"))


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--root", default=os.environ.get("KITTI_ROOT", ""))
    ap.add_argument("--seq", default="05")
    ap.add_argument("--frames", type=int, default=0, help="0 = the whole sequence")
    ap.add_argument("--out", default="")
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--variant", type=int, default=2, help="parameter set: 2 = subMapOptimization (default), 3 = loop-closure copy")
    ap.add_argument("--mode", choices=("submap", "odom"), default="submap",
                    help="submap: labelled sweeps through the sliding local map (copy #2); odom: raw sweeps, keyframe odometry (copy #1)")
    ap.add_argument("--check-oracle", type=int, default=0, metavar="N", help="also run the CPU restatement on the first N frames")
    ap.add_argument("--exact", action="store_true",
                    help="register with the exact-arithmetic build (the reference's arithmetic operation for operation, ~0.5x the speed): "
                         "with --check-oracle the poses must then equal the restatement's to the bit (reported as frames_bit_identical)")
    args = ap.parse_args(argv)
    seq_dir = os.path.join(args.root, "sequences", args.seq, "velodyne")
    if not args.root or not os.path.isdir(seq_dir):
        print(f"kitti_replay: {seq_dir!r} not found — mount the dataset and pass --root or set KITTI_ROOT", file=sys.stderr)
        return 2
    import numpy as np
    import lisreg
    from lisreg import replay
    odom = args.mode == "odom"
    frames = (replay.kitti_raw_sequence if odom else replay.kitti_sequence)(args.root, args.seq, args.frames or None)
    ctx = lisreg.Context(args.device)
    if args.exact:
        ctx.set_option("exact_arithmetic", 1)
    # the device-resident drivers: the sweep goes up once, every cloud of the frame stays in HBM (bit-identical to the host-cloud drivers)
    r = replay.DeviceOdomReplayer(ctx) if odom else replay.DeviceReplayer(ctx, args.variant)
    recs, kept = [], []
    t0 = time.perf_counter()
    for cloud, _ in frames:
        if len(recs) < args.check_oracle:
            kept.append(cloud)
        rec = r.step(cloud)
        recs.append(rec)
        if rec["stats"] and rec["stats"]["status"] != 0:
            print(f"frame {rec['frame']}: status {rec['stats']['status']} (too few features / correspondences), pose kept", file=sys.stderr)
    dt = time.perf_counter() - t0
    ctx.close()
    if args.out:
        replay.write_trajectory(args.out, [rec["T"] for rec in recs])
    summary = dict(sequence=args.seq, mode=args.mode, frames=len(recs), seconds=round(dt, 3), frames_per_s=round(len(recs) / max(dt, 1e-9), 2),
                   mean_iters=float(np.mean([rec["stats"]["iters"] for rec in recs if rec["stats"]] or [0])),
                   final_pose=[float(v) for v in recs[-1]["T"]] if recs else None, trajectory=args.out or None)
    if args.check_oracle > 0 and kept:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import replay_oracle as ro                       # checker only
        ref = (ro.replay_odom if odom else ro.replay)(kept, n_threads=min(16, os.cpu_count() or 1))
        d = [np.abs(np.asarray(a["T"], np.float64) - np.asarray(b["T"], np.float64)) for a, b in zip(recs, ref)]
        summary["oracle_check"] = dict(frames=len(ref), max_rot_diff_rad=float(max(x[:3].max() for x in d)),
                                       max_trans_diff_m=float(max(x[3:].max() for x in d)),
                                       frames_bit_identical=int(sum(np.array_equal(np.asarray(a["T"], np.float32), np.asarray(b["T"], np.float32))
                                                                    for a, b in zip(recs, ref))))
    print(json.dumps(summary))
    return 0


if __name__ == "__main__":
    sys.exit(main())
