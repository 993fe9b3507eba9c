"""Evidence run (not a test): lisreg_extract_features against the CPU restatement on random raw sweeps of several shapes, scan-ordered and
shuffled, thresholds varied — the five output clouds must hold the same points in the same order.   python tests/feature_sweep.py [n]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "lis-slam_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import lisreg
import oracle_ctypes as oc
from lisreg import synth
from test_features import NAMES, same_points
oc.build()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
ctx = lisreg.Context(0)
rng = np.random.default_rng(99)
shapes = [(16, 450), (16, 900), (32, 1024), (64, 1800), (64, 900), (128, 2048), (40, 1200)]
bad = []; pts = 0
for k in range(n):
    h, w = shapes[k % len(shapes)]
    rate = int(rng.choice([1, 1, 2, 4])); shuffle = bool(rng.integers(0, 2))
    edge, surf = float(rng.choice([1.0, 0.5, 2.0])), float(rng.choice([0.1, 0.05, 0.2]))
    c = synth.make_raw_scan(h, w, 9000 + k, shuffle=shuffle)
    ro = oc.extract_features(c, oc.FeatureParams(h, w, rate, 0.0, 70.0, edge, surf))
    rg = ctx.extract_features(c, lisreg.FeatureParams(h, w, rate, 0.0, 70.0, edge, surf))
    ok = all(len(rg[q]) == len(ro[q]) and same_points(rg[q], c[ro[q]]) for q in NAMES)
    pts += len(c)
    if not ok:
        bad.append(k); print(f"case {k}: {h}x{w} rate {rate} shuffle {shuffle} thresholds {edge}/{surf}: DIFFERS", {q: (len(rg[q]), len(ro[q])) for q in NAMES})
print(f"== {n - len(bad)} of {n} sweeps: the five feature clouds equal the oracle's (same points, same order); {pts} input points; differing cases: {bad}")
ctx.close()
