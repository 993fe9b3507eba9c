"""probe (GPU box): distribution of the entry counts of the cell rows of the configs[1] submap — how much of a 64-entry row is padding.
python tests/probes/crow_counts.py"""
import sys, os, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "lis-slam_amd"))
import lisreg
from lisreg import synth
ctx = lisreg.Context(0)
tc, ts = synth.make_submap(200000, 42)
ctx.set_target(tc, ts)
for kind in (0, 1):
    g = ctx.target_cell_rows(0, kind)
    cnt = np.asarray(g["count"]); R = g["n_rows"]
    written = np.maximum(4, (cnt + 3) & ~3)
    print(f"kind {kind}: rows {R}, mean count {cnt.mean():.1f}, mean written entries (groups of 4, >= 4) {written.mean():.1f} of {g['k']}, "
          f"histogram (0, 1-4, 5-8, 9-16, 17-32, 33-48, 49-63, 64): "
          f"{[int(((cnt >= a) & (cnt <= b)).sum()) for a, b in ((0, 0), (1, 4), (5, 8), (9, 16), (17, 32), (33, 48), (49, 63), (64, 64))]}")
