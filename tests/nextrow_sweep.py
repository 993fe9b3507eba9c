"""Evidence run (not a test): the map-maintenance primitives and the voxel grid against the CPU restatement on random inputs — cloud sizes
from one point to 200 k, clustered and uniform, leaf sizes 0.05 .. 2 m, search caps, random boxes.   python tests/nextrow_sweep.py [n]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "lis-slam_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import lisreg
import oracle_ctypes as oc
from lisreg import synth
oc.build()
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
ctx = lisreg.Context(0)
rng = np.random.default_rng(2024)
FIELDS = ("x", "y", "z", "intensity", "label")


def cloud(n):
    kind = rng.integers(0, 3)
    if kind == 0: xyz = rng.uniform(-40, 40, (n, 3)) * np.array([1, 1, 0.1])
    elif kind == 1: xyz = rng.normal(0, 1, (n, 3)) * rng.uniform(0.05, 15) + rng.uniform(-20, 20, 3)              # one blob: crowded voxels
    else: xyz = np.concatenate([rng.normal(0, 0.3, (n - n // 2, 3)) + rng.uniform(-30, 30, 3), rng.uniform(-50, 50, (n // 2, 3))])
    c = synth.to_pcl(xyz.astype(np.float32), rng.integers(0, 20, n).astype(np.uint16))
    c["intensity"] = rng.uniform(0, 200, n).astype(np.float32)
    return c


def same(a, b):
    return len(a) == len(b) and all(np.array_equal(a[f], b[f]) for f in FIELDS)


bad = []; tally = dict(voxel=0, nearest=0, dynamic=0, bbx=0)
for k in range(n_cases):
    n_map = int(rng.choice([1, 4, 5, 100, 5000, 60000, 200000])); n_q = int(rng.choice([1, 7, 3000, 50000]))
    m, q = cloud(n_map), cloud(n_q)
    ok = True
    leaf = float(rng.choice([0.05, 0.2, 0.4, 1.0, 2.0]))
    so, wo = oc.voxel_grid(m, leaf); sg, wg = ctx.voxel_downsample(m, leaf)
    ok &= so == sg and (so != 0 or same(wg, wo)); tally["voxel"] += 1
    ctx.map_index_set(7, m)
    cap = float(rng.choice([0.5, 3.0, 1e18]))
    io, do = oc.nearest(m, q, cap); ig, dg = ctx.nearest(7, q, cap)
    ok &= np.array_equal(ig, io) and np.array_equal(dg[ig >= 0], do[io >= 0]); tally["nearest"] += 1
    args = (float(rng.choice([5.0, 30.0, 100.0])), float(rng.choice([0.1, 0.3, 3.4028234663852886e38])), float(rng.choice([1.0, 3.0, 3.4028234663852886e38])), float(rng.choice([0.0, 0.03])))
    ko, _ = oc.dynamic_filter(m, q, *args); kg, _ = ctx.dynamic_filter(7, q, *args)
    ok &= same(kg, ko); tally["dynamic"] += 1
    lo = rng.uniform(-40, 0, 3); box = np.concatenate([lo, lo + rng.uniform(0, 60, 3)])
    dele = bool(rng.integers(0, 2))
    ok &= same(ctx.bbx_filter(m, box, dele), oc.bbx_filter(m, box, dele)); tally["bbx"] += 1
    if not ok:
        bad.append(k); print(f"case {k}: map {n_map} query {n_q} leaf {leaf} cap {cap}: DIFFERS")
print(f"== {n_cases - len(bad)} of {n_cases} random cases: voxel grid, nearest (index and squared distance), dynamic_filter and bbx_filter equal the oracle's, "
      f"field by field; differing cases: {bad}")
ctx.close()
