#!/usr/bin/env python3
"""Evidence run (GPU): random loop-closure candidate batches through lisreg_icp_align_batch against the same candidates through
lisreg_icp_align one by one (independent and chained prev_mse), every result field compared bit for bit; a sample against the oracle.
usage: python tests/icp_batch_sweep.py [n_batches] > profiles/r04_icp_batch_sweep.txt"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lis-slam_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import lisreg
from lisreg import synth
import oracle_ctypes as oc
from test_icp import _case, _same, _pose_diff

oc.build()
n_batches = int(sys.argv[1]) if len(sys.argv) > 1 else 60
ctx = lisreg.Context(0)
pool = [_case(900 + k, n_map=int(20000 + 15000 * (k % 4)), trans=0.2 + 0.25 * (k % 5), rot_deg=1.0 + (k % 4), hw=((16, 450), (32, 900))[k % 2]) for k in range(8)]
for k, (tgt, _, _) in enumerate(pool):
    ctx.map_index_set(40 + k, tgt)
tot = eq_ind = eq_chain = 0
orc_checked = orc_ok = 0
states = {}
for b in range(n_batches):
    rng = np.random.default_rng(10_000 + b)
    n = int(rng.integers(1, 24))
    items = []
    for _ in range(n):
        k = int(rng.integers(0, 8))
        src = pool[k][1]
        kind = rng.integers(0, 10)
        if kind == 0:
            src = src[: int(rng.integers(0, 3))]                       # 0-2 points
        elif kind == 1:
            src = src.copy(); src["x"] += 500.0                        # nothing within reach
        elif kind == 2:
            src = src[:: int(rng.integers(2, 9))]                      # thinned
        g = None
        if rng.uniform() < 0.4:
            g = synth.pose_matrix([*rng.normal(0, 0.01, 3), *rng.normal(0, 0.1, 3)]).astype(np.float32)
        items.append((40 + k, np.ascontiguousarray(src), g))
    kindp = int(rng.integers(0, 2))
    p = lisreg.icp_default_params(kindp)
    if rng.uniform() < 0.3:
        p.max_iters = int(rng.integers(1, 12))
    rb = ctx.icp_align_batch(items, p, chain_prev_mse=False)
    rc = ctx.icp_align_batch(items, p, chain_prev_mse=True)
    ps = lisreg.icp_default_params(kindp); ps.max_iters = p.max_iters
    for j, (slot, src, g) in enumerate(items):
        r1 = ctx.icp_align(slot, src, p, guess=g)
        r2 = ctx.icp_align(slot, src, ps, guess=g)
        ps.prev_mse = r2["prev_mse"]
        tot += 1
        eq_ind += int(_same(rb[j], r1)); eq_chain += int(_same(rc[j], r2))
        states[r1["state"]] = states.get(r1["state"], 0) + 1
        if not _same(rb[j], r1) or not _same(rc[j], r2):
            print("MISMATCH batch", b, "item", j, rb[j], r1, rc[j], r2)
    if b % 10 == 0:                                                    # one candidate of every tenth batch against the oracle
        slot, src, g = items[0]
        if len(src) > 2 and rb[0]["state"] != lisreg.ICP_NO_CORRESPONDENCES:
            po = oc.icp_default_params(kindp); po.max_iters = p.max_iters
            ro_ = oc.icp_align(pool[slot - 40][0], src, po, guess=g)
            dr, dt = _pose_diff(rb[0]["T"], ro_["T"])
            orc_checked += 1
            orc_ok += int(rb[0]["state"] == ro_["state"] and abs(rb[0]["iters"] - ro_["iters"]) <= 1 and dr < 1e-3 and dt < 1e-3)
ctx.close()
print(f"{n_batches} random batches (1-23 candidates: 8 targets of 20-65 k points, sources of 0 ... 28 k points, guesses, both parameter sets, "
      f"max_iters 1 ... 50), {tot} candidates")
print(f"  independent form: {eq_ind} of {tot} equal lisreg_icp_align on the candidate alone in EVERY field, bit for bit")
print(f"  chained prev_mse: {eq_chain} of {tot} equal a sequential loop of lisreg_icp_align calls feeding prev_mse forward, bit for bit")
print(f"  convergence states met: {dict(sorted(states.items()))}  (1 iterations, 2 transform, 3 abs MSE, 4 rel MSE, 5 no correspondences)")
print(f"  against the oracle: {orc_ok} of {orc_checked} sampled candidates with the same state, iterations within one, transform within 1e-3")
