"""Ad-hoc GPU debug run (not a test): prints oracle-vs-GPU details for one case."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "lis-slam_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import lisreg, oracle_ctypes as oc
from lisreg import synth
from helpers import copy_params, pose_err
oc.build()
ctx = lisreg.Context(0)
for variant, labelled, seed in ((1, False, 1000), (2, True, 1002)):
    case = synth.make_case(h=16, w=450, m_points=20000, scan_seed=seed, labelled=labelled)
    po = oc.default_params(variant); pg = copy_params(po, lisreg.Params)
    To, so, tro = oc.align(case["tgt_corner"], case["tgt_surf"], case["src_corner"], case["src_surf"], case["T_init"], po)
    ctx.set_target(case["tgt_corner"], case["tgt_surf"])
    t = time.time()
    Tg, sg, trg = ctx.align(case["src_corner"], case["src_surf"], case["T_init"], pg)
    print("gpu align wall", time.time() - t)
    print("oracle", To, so); print("gpu   ", Tg, sg); print("err", pose_err(Tg, To))
    for k in range(min(len(tro), len(trg))):
        print(k, tro[k, 0], trg[k, 0], pose_err(trg[k, 49:55], tro[k, 49:55]), np.abs(trg[k,1:37]-tro[k,1:37]).max()/np.abs(tro[k,1:37]).max())
