"""Ad-hoc (not a test): print oracle vs GPU traces for one random-sweep seed."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("lis-slam_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np
import lisreg, oracle_ctypes as oc
from lisreg import synth
from helpers import copy_params, pose_err
oc.build()
for seed in [int(a) for a in sys.argv[1:]] or range(12):
    rng = np.random.default_rng(9000 + seed)
    variant = int(rng.integers(1, 4)); labelled = variant != 1 and bool(rng.integers(0, 2))
    h, w = int(rng.choice([8, 16, 32])), int(rng.choice([225, 450, 900])); m_points = int(rng.choice([15000, 30000, 60000]))
    pose_xy = [None, (30.0, -28.0), (-35.0, 10.0), (5.0, 36.0)][int(rng.integers(0, 4))]
    case = synth.make_case(h=h, w=w, m_points=m_points, scan_seed=9100 + seed, labelled=labelled,
                           trans=float(rng.uniform(0.05, 0.5)), rot_deg=float(rng.uniform(0.2, 3.0)), pose_xy=pose_xy)
    fixed = int(rng.choice([0, 0, 4, 12])); imu = None if rng.integers(0, 2) else (1, float(rng.uniform(-0.05, 0.05)), float(rng.uniform(-0.05, 0.05)))
    po = oc.default_params(variant); po.fixed_iters = fixed; pg = copy_params(po, lisreg.Params)
    To, so, tro = oc.align(case["tgt_corner"], case["tgt_surf"], case["src_corner"], case["src_surf"], case["T_init"], po, oc.Imu(*imu) if imu else None)
    c = lisreg.Context(0); c.set_target(case["tgt_corner"], case["tgt_surf"])
    Tg, sg, trg = c.align(case["src_corner"], case["src_surf"], case["T_init"], pg, lisreg.Imu(*imu) if imu else None); c.close()
    print(f"seed {seed} variant {variant} labelled {labelled} fixed {fixed} conv ({po.conv_deg:.4f},{po.conv_cm:.3f}) sizes {len(case['src_corner'])}/{len(case['src_surf'])} vs {len(case['tgt_corner'])}/{len(case['tgt_surf'])}")
    print("  oracle", so["iters"], so["deltaR"], so["deltaT"], " gpu", sg["iters"], sg["deltaR"], sg["deltaT"], " final pose err", pose_err(Tg, To))
    for k in range(max(len(tro), len(trg))):
        def d(tr):
            if k >= len(tr): return "   -"
            X = tr[k, 43:49]; return "n=%5d dR=%.5f dT=%.5f" % (tr[k, 0], np.linalg.norm(X[:3] * 57.29578), np.linalg.norm(X[3:] * 100))
        e = pose_err(trg[k, 49:55], tro[k, 49:55]) if k < min(len(tro), len(trg)) else None
        print("   it", k, "| oracle", d(tro), "| gpu", d(trg), "| pose diff", e)
