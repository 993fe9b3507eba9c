"""Generates tests/golden/*.npz: seeded inputs + expected per-iteration traces of the registration path.

Expected values come from oracle/lisreg_numpy.py — the independent numpy/LAPACK/cKDTree mirror of
/root/reference/src/node/odomEstimationNode.cpp:596-974 (and the label-weighted copies) — NOT from the C oracle
or the HIP library, which are both checked AGAINST these files.  The reference itself has no golden vectors and
cannot run here, so these fixtures are the committed contract ("parity unpinned" upstream).
Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "lis-slam_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
from lisreg import synth          # noqa: E402
import lisreg_numpy as ln         # noqa: E402
import oracle_ctypes as oc        # noqa: E402  (only for the literal parameter sets)

HERE = os.path.dirname(os.path.abspath(__file__))


def pdict(variant, fixed_iters=0):
    p = oc.default_params(variant)
    p.fixed_iters = fixed_iters
    d = {f: getattr(p, f) for f, _ in p._fields_ if f != "label_score"}
    d["label_score"] = [float(x) for x in p.label_score]
    return d


def run(name, case, variant, fixed_iters=0, degenerate_in=0):
    p = pdict(variant, fixed_iters)
    x = synth.pcl_xyz
    T, st, tr = ln.align(x(case["tgt_corner"]), x(case["tgt_surf"]), x(case["src_corner"]), x(case["src_surf"]),
                         case["src_corner"]["label"], case["src_surf"]["label"], case["T_init"], p, degenerate_in)
    n = len(tr)
    trace = np.zeros((n, 56), np.float32)
    for k, r in enumerate(tr):
        trace[k, 0] = r["n_corr"]
        trace[k, 49:55] = r["T"]
        if r["solved"]:
            trace[k, 1:37] = r["AtA"].ravel(); trace[k, 37:43] = r["AtB"]; trace[k, 43:49] = r["X"]; trace[k, 55] = 1
    def pack(c):
        return np.concatenate([synth.pcl_xyz(c), c["label"].astype(np.float32)[:, None]], 1).astype(np.float32)
    np.savez_compressed(os.path.join(HERE, name + ".npz"),
                        tgt_corner=pack(case["tgt_corner"]), tgt_surf=pack(case["tgt_surf"]),
                        src_corner=pack(case["src_corner"]), src_surf=pack(case["src_surf"]),
                        T_init=case["T_init"].astype(np.float32), T_true=case["T_true"].astype(np.float32),
                        variant=np.int32(variant), fixed_iters=np.int32(fixed_iters), degenerate_in=np.int32(degenerate_in),
                        T_expected=T.astype(np.float32), trace=trace,
                        stats=np.array([st["iters"], st["degenerate"], st["n_corr_last"], st["status"]], np.int32),
                        deltas=np.array([st["deltaR"], st["deltaT"]], np.float32))
    print(name, "iters", st["iters"], "deg", st["degenerate"], "n_corr", st["n_corr_last"], "status", st["status"],
          "T", T, "| sizes", len(case["src_corner"]), len(case["src_surf"]), len(case["tgt_corner"]), len(case["tgt_surf"]))


if __name__ == "__main__":
    kw = dict(h=16, w=300, m_points=150000, local_radius=9.0)
    run("odom_2001", synth.make_case(scan_seed=2001, **kw), 1)
    run("keyframe_2002", synth.make_case(scan_seed=2002, labelled=True, **kw), 2)                 # open floor: degenerate
    run("keyframe_2005_corner", synth.make_case(scan_seed=2005, labelled=True, pose_xy=(33.0, 31.0), **kw), 2)
    c3 = synth.make_case(scan_seed=2003, labelled=True, **kw)
    c3["tgt_corner"] = c3["tgt_corner"][:0]                     # variant #3 skips the empty corner stage (:4505)
    run("submap_2003_nocorner", c3, 3)
    c6 = synth.make_case(scan_seed=2006, labelled=True, pose_xy=(-32.0, 34.0), **kw)
    run("submap_2006_corner", c6, 3)
    run("odom_2004_fixed10", synth.make_case(scan_seed=2004, **kw), 1, fixed_iters=10)
    run("plane_degenerate", synth.make_plane_case(), 1)
