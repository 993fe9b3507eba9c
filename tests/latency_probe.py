"""Ad-hoc (not a test): per-call latency of lisreg_align for a realistic odometry-sized problem."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lis-slam_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import lisreg, oracle_ctypes as oc
from lisreg import synth
ctx = lisreg.Context(0)
case = synth.make_case(h=32, w=900, m_points=50000, scan_seed=1000)
# voxel-ish thinning of the source to the reference's typical sizes (N_c ~ 1-4k, N_p ~ 5-15k)
sc, ss = case["src_corner"], case["src_surf"][::2]
print("sizes", len(sc), len(ss), len(case["tgt_corner"]), len(case["tgt_surf"]))
p = lisreg.default_params(1)
for rep in range(3):
    t0 = time.perf_counter(); ctx.set_target(case["tgt_corner"], case["tgt_surf"]); t1 = time.perf_counter()
    T, st, tr = ctx.align(sc, ss, case["T_init"], p); t2 = time.perf_counter()
    print(f"set_target {1e3*(t1-t0):.3f} ms  align {1e3*(t2-t1):.3f} ms  iters {st['iters']}")
ts = []
for rep in range(20):
    t1 = time.perf_counter(); T, st, tr = ctx.align(sc, ss, case["T_init"], p); ts.append(time.perf_counter() - t1)
print("align only: median %.3f ms  min %.3f ms" % (1e3*np.median(ts), 1e3*min(ts)))
ctx.set_profiling(True); T, st, tr = ctx.align(sc, ss, case["T_init"], p); print(ctx.timing())
po = oc.default_params(1)
t0 = time.perf_counter(); To, so, _ = oc.align(case["tgt_corner"], case["tgt_surf"], sc, ss, case["T_init"], po); print("oracle 1 thread: %.3f ms" % (1e3*(time.perf_counter()-t0)), so["iters"], np.abs(To-T).max())
