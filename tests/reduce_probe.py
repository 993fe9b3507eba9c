"""Ad-hoc measurement (not a test): final poses of a BASELINE configs[1]-shaped batch (16 scans vs the 200 k submap, 10 fixed GN
iterations), saved for a diff between two builds of the library (e.g. the default fp32-in-wave reduction vs -DLISREG_REDUCE_FP64)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lis-slam_amd"))
import numpy as np, lisreg
from lisreg import synth
tc, ts = synth.make_submap(200000)
cases, T0 = [], []
for i in range(16):
    sc = synth.make_scan(64, 1800, 1000 + i)
    cases.append(dict(src_corner=sc["corner"], src_surf=sc["surf"]))
    T0.append(synth.perturb_pose(sc["T_true"], np.random.default_rng(1000 + i + 7919)))
p = lisreg.default_params(1); p.fixed_iters = 10
ctx = lisreg.Context(0)
ctx.set_option("search_mode", 3)
ctx.set_target(tc, ts)
T, st = ctx.align_batch(cases, np.array(T0, np.float32), p)
out = os.path.join(ROOT, "gpurun_out", sys.argv[1] if len(sys.argv) > 1 else "poses.npy")
np.save(out, T); print("saved", out, T.shape, st[0])
