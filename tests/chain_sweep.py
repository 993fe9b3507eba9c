"""Evidence run (not a test): both sequential chains with the exact-arithmetic build against their oracle chains on full-size drives —
every frame's pose must be the oracle's to the bit.   python tests/chain_sweep.py [n_frames]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "lis-slam_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import lisreg
import oracle_ctypes as oc
import replay_oracle as ro
from lisreg import replay
oc.build()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
for seed in (7, 21):
    frames = [c for c, _ in replay.synthetic_drive(n, seed=seed)]
    ref = ro.replay(frames, n_threads=16)
    ctx = lisreg.Context(0); ctx.set_option("exact_arithmetic", 1)
    got = replay.replay(ctx, frames, device_resident=True); ctx.close()
    same = sum(np.array_equal(np.asarray(a["T"], np.float32), np.asarray(b["T"], np.float32)) for a, b in zip(got, ref))
    it = all(a["stats"]["iters"] == b["stats"]["iters"] and a["stats"]["n_corr_last"] == b["stats"]["n_corr_last"] for a, b in zip(got, ref) if a["stats"])
    print(f"frame loop (configs[2]), 64x1800 labelled drive seed {seed}: poses bit-identical in {same} of {n} frames; iteration and correspondence counts equal: {it}; "
          f"last frame: {got[-1]['stats']['n_corr_last']} correspondences against {got[-1]['n_target_surf']} + {got[-1]['n_target_corner']} target points")
for seed in (11, 23):
    sweeps = [c for c, _ in replay.synthetic_raw_drive(n, seed=seed)]
    ref = ro.replay_odom(sweeps, n_threads=16)
    ctx = lisreg.Context(0); ctx.set_option("exact_arithmetic", 1)
    got = replay.replay_odom(ctx, sweeps, device_resident=True); ctx.close()
    same = sum(np.array_equal(np.asarray(a["T"], np.float32), np.asarray(b["T"], np.float32)) for a, b in zip(got, ref))
    keys = all(a["keyframe"] == b["keyframe"] for a, b in zip(got, ref))
    print(f"odometry loop (configs[0]), 64x1800 raw drive seed {seed}: poses bit-identical in {same} of {n} frames; key-frame decisions equal: {keys} ({sum(a['keyframe'] for a in got)} key frames)")
