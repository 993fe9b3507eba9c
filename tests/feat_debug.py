"""Ad-hoc (not a test): locate the first difference between oracle and HIP feature lists."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("lis-slam_amd", "oracle"): sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np, lisreg, oracle_ctypes as oc
from lisreg import synth
oc.build()
h, w, rate = 16, 450, 1
c = synth.make_raw_scan(h, w, 7211)
po = oc.FeatureParams(h, w, rate, 0.0, 70.0, 1.0, 0.1); pg = lisreg.FeatureParams(h, w, rate, 0.0, 70.0, 1.0, 0.1)
ro = oc.extract_features(c, po); ctx = lisreg.Context(0); rg = ctx.extract_features(c, pg)
key = {tuple(np.frombuffer(c[i].tobytes(), np.uint8)): i for i in range(len(c))}
for k in ("corner", "surface", "corner_sharp", "surface_sharp"):
    gi = np.array([key[tuple(np.frombuffer(r.tobytes(), np.uint8))] for r in rg[k]])
    oi = ro[k]
    same_set = set(gi) == set(oi)
    d = np.nonzero(gi[:min(len(gi), len(oi))] != oi[:min(len(gi), len(oi))])[0]
    print(k, len(gi), len(oi), "same set" if same_set else f"sets differ: only gpu {sorted(set(gi)-set(oi))[:5]} only orc {sorted(set(oi)-set(gi))[:5]}",
          "first diff at", d[:5], "gpu", gi[d[:5]], "orc", oi[d[:5]], "rings", c["ring"][gi[d[:3]]], c["ring"][oi[d[:3]]])
