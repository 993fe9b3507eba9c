"""diagnostic: python tests/diag_frontend_seed.py <seed> — where the cell rows' neighbours differ from the walk's for one sweep configuration"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "lis-slam_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import lisreg
import oracle_ctypes as oc
from lisreg import synth
from test_exact import sweep_case
from helpers import copy_params
seed = int(sys.argv[1])
case, variant, fixed, imu = sweep_case(seed)
p = copy_params(oc.default_params(variant), lisreg.Params)
p.fixed_iters = fixed if fixed > 0 else 6
out = {}
for mode, lanes, ties in ((1, 1, 1), (5, 1, 1), (5, 1, 0), (1, 1, 0)):
    c = lisreg.Context(0)
    c.set_option("canonical_ties", ties); c.set_option("search_mode", mode); c.set_option("lanes_per_query", lanes); c.set_option("dump_neighbors", 1)
    c.set_target(case["tgt_corner"], case["tgt_surf"])
    T, st, tr = c.align(case["src_corner"], case["src_surf"], case["T_init"], p, lisreg.Imu(*imu) if imu else None)
    nb = c.neighbors(len(case["src_corner"]) + len(case["src_surf"]))
    out[(mode, ties)] = (T, tr, nb, st)
    if mode == 5 and ties == 1:
        rows = [c.target_cell_rows(0, k) for k in (0, 1)]
        idx = [c.target_index(0, k) for k in (0, 1)]
    c.close()
a, b = out[(1, 1)], out[(5, 1)]
print("iters", a[3]["iters"], b[3]["iters"], "poses equal", np.array_equal(a[0], b[0]))
nc = len(case["src_corner"])
diff = np.flatnonzero((a[2][:5] != b[2][:5]).any(0))
print("differing queries:", diff, "of", a[2].shape[1], "(corner queries:", nc, ")")
src = np.concatenate([synth.pcl_xyz(case["src_corner"]), synth.pcl_xyz(case["src_surf"])]).astype(np.float64)
for qi in diff[:5]:
    kind = 0 if qi < nc else 1
    tgt = synth.pcl_xyz(case["tgt_corner"] if kind == 0 else case["tgt_surf"]).astype(np.float64)
    # pose the last executed iteration searched with
    k_last = len(a[1]) - 1
    T_it = a[1][k_last, 49:55].astype(np.float64)
    M = synth.pose_matrix(T_it)
    q = M[:3, :3] @ src[qi] + M[:3, 3]
    for name, r in (("walk", a), ("rows", b)):
        ids = r[2][:5, qi]
        d = [float(np.sum((tgt[i] - q) ** 2)) if i >= 0 else None for i in ids]
        print(f"  query {qi} kind {kind} {name}: ids {ids.tolist()} d2 {d} flag {r[2][5, qi]}")
    d_all = np.sum((tgt - q) ** 2, 1)
    o = np.argsort(d_all)[:7]
    print("    float64 nearest:", o.tolist(), d_all[o].tolist())
print("rows (ties off) vs walk (ties off) differing:", int((out[(5, 0)][2][:5] != out[(1, 0)][2][:5]).any(0).sum()))
# tables of both targets, for comparison between two libraries (python tests/diag_frontend_seed.py <seed> <tag> writes gpurun_out/diag_<tag>.npz)
if len(sys.argv) > 2:
    np.savez(os.path.join(ROOT, "gpurun_out", f"diag_{sys.argv[2]}.npz"), tab0=rows[0]["table"], tab1=rows[1]["table"], ids1=rows[1]["ids"], cnt1=rows[1]["count"], rho1=rows[1]["rho2"],
             nb_rows=b[2], nb_walk=a[2])
    print("saved", sys.argv[2], "rows", rows[0]["n_rows"], rows[1]["n_rows"])
