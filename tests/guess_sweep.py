#!/usr/bin/env python3
"""Evidence run (CPU only): lisreg_update_initial_guess against the branch-by-branch restatement (oracle/replay_oracle.py) on random
availability sequences, both node copies — poses and transPredictionMapped compared bit for bit after every call.
usage: python tests/guess_sweep.py [n_sequences] > profiles/r04_guess_sweep.txt"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lis-slam_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import lisreg
import replay_oracle as ro

n_seq = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
calls = equal = 0
branch = {"first": 0, "odom_first": 0, "odom_incr": 0, "imu_incr": 0, "const_vel": 0, "none": 0}
for seq in range(n_seq):
    rng = np.random.default_rng(seq)
    variant, heading = seq & 1, bool((seq >> 1) & 1)
    g, o = lisreg.InitialGuess(variant, heading), ro.InitialGuessOracle(variant, heading)
    T = np.zeros(6, np.float32)
    imu = rng.uniform(-0.1, 0.1, 3); imu[2] = rng.uniform(-3.1, 3.1)
    odo = np.concatenate([rng.uniform(-100, 100, 3), rng.uniform(-0.1, 0.1, 2), rng.uniform(-3.1, 3.1, 1)])
    p_odom, p_imu = rng.uniform(0, 1), rng.uniform(0, 1)
    for frame in range(int(rng.integers(3, 40))):
        imu = imu + rng.normal(0, 0.01, 3)
        odo = odo + np.concatenate([rng.normal(0.5, 0.3, 3), rng.normal(0, 0.01, 3)])
        oa, ia = bool(rng.uniform() < p_odom), bool(rng.uniform() < p_imu)
        Tg, pg = g.update(T, oa, ia, imu, odo)
        To, po = o.update(T, oa, ia, imu, odo)
        calls += 1
        same = np.array_equal(Tg, To) and (pg is None) == (po is None) and (pg is None or np.array_equal(pg, po))
        equal += int(same)
        if not same:
            print("MISMATCH", seq, frame, variant, oa, ia, Tg, To)
        T = (Tg + np.concatenate([rng.normal(0, 2e-3, 3), rng.normal(0, 0.05, 3)])).astype(np.float32)
print(f"{n_seq} sequences, {calls} calls of updateInitialGuess (both node copies, random odomAvailable / imuAvailable per frame): "
      f"{equal} of {calls} equal to the restatement bit for bit (pose and transPredictionMapped)")
