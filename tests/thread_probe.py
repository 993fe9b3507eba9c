"""Ad-hoc (not a test): where does time go with two contexts on two host threads?"""
import os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lis-slam_amd"))
import numpy as np
import lisreg
from lisreg import synth
case = synth.make_case(h=16, w=450, m_points=20000, scan_seed=1400, labelled=True)
p = lisreg.default_params(2)
def work(tag, reps):
    t0 = time.perf_counter(); ctx = lisreg.Context(0); t1 = time.perf_counter()
    ts = []
    for _ in range(reps):
        a = time.perf_counter(); ctx.set_target(case["tgt_corner"], case["tgt_surf"]); b = time.perf_counter()
        ctx.align(case["src_corner"], case["src_surf"], case["T_init"], p); c = time.perf_counter()
        ts.append((b - a, c - b))
    t2 = time.perf_counter(); ctx.close(); t3 = time.perf_counter()
    print(f"{tag}: create {1e3*(t1-t0):.1f} ms, close {1e3*(t3-t2):.1f} ms, per rep (set_target, align) ms:",
          [(round(1e3*x,2), round(1e3*y,2)) for x, y in ts[:3]], "...", [(round(1e3*x,2), round(1e3*y,2)) for x, y in ts[-2:]], flush=True)
work("warm-up single", 4)
work("single again", 4)
t = time.perf_counter()
th = [threading.Thread(target=work, args=(f"thread{k}", 8)) for k in range(2)]
[x.start() for x in th]; [x.join() for x in th]
print("two threads total %.1f ms" % (1e3*(time.perf_counter()-t)))
