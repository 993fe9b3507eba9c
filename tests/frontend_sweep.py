"""Evidence run (not a test): with "canonical_ties" set, the search front-ends of the PRODUCTION build — cell walk with eight lanes per
query, cell walk with one, graph scan, cell rows — must return the same five neighbours in the same order for every query of every iteration, hence
bit-identical poses.  Random configurations of tests/test_exact.py::sweep_case.   python tests/frontend_sweep.py [first_seed] [n]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "lis-slam_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import lisreg
import oracle_ctypes as oc
from test_exact import sweep_case
from helpers import copy_params
first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
n = int(sys.argv[2]) if len(sys.argv) > 2 else 100
bad = []; queries = 0
for seed in range(first, first + n):
    case, variant, fixed, imu = sweep_case(seed)
    p = copy_params(oc.default_params(variant), lisreg.Params)
    p.fixed_iters = fixed if fixed > 0 else 6
    runs = []
    for mode, lanes in ((1, 8), (1, 1), (3, 1), (5, 1)):
        c = lisreg.Context(0)
        c.set_option("canonical_ties", 1); c.set_option("search_mode", mode); c.set_option("lanes_per_query", lanes); c.set_option("dump_neighbors", 1)
        c.set_target(case["tgt_corner"], case["tgt_surf"])
        T, st, tr = c.align(case["src_corner"], case["src_surf"], case["T_init"], p, lisreg.Imu(*imu) if imu else None)
        nb = c.neighbors(len(case["src_corner"]) + len(case["src_surf"]))
        runs.append((T, tr, nb)); c.close()
    ok = all(np.array_equal(runs[0][0], r[0]) and np.array_equal(runs[0][1], r[1]) and np.array_equal(runs[0][2][:6], r[2][:6]) for r in runs[1:])
    queries += runs[0][2].shape[1]
    if not ok:
        bad.append(seed)
        d = [int((runs[0][2][:5] != r[2][:5]).any(0).sum()) for r in runs[1:]]
        print(f"seed {seed}: front-ends DIFFER — queries with other neighbours than the eight-lane walk: one-lane walk {d[0]}, graph scan {d[1]}, cell rows {d[2]}")
print(f"== {n - len(bad)} of {n} configurations: the four front-ends (walk x 8 lanes, walk, graph scan, cell rows) bit-identical (poses and traces of every iteration, neighbours and accept flags of the last); "
      f"{queries} queries; differing seeds: {bad}")
