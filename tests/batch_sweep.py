"""Evidence run (not a test): a frame registers identically alone and inside any batch.  Random batches (3-12 registrations, two target
slots, mixed scan sizes) through lisreg_align_batch with the search front-end left at auto, against the same registrations one by one
(which take the eight-lane walk): with "canonical_ties" every pose, statistic and trace must be the same bits.
   python tests/batch_sweep.py [n_batches]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "lis-slam_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import lisreg
from lisreg import synth
n_b = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(4242)
ctx = lisreg.Context(0)
ctx.set_option("canonical_ties", 1)
bad = []; regs = 0; fronts = {}
for b in range(n_b):
    variant = int(rng.choice([1, 2]))
    labelled = variant == 2
    tgts = [synth.make_submap(int(rng.choice([8000, 30000, 60000])), 500 + 2 * b + s, labelled=labelled) for s in range(2)]
    for s, (tc, ts) in enumerate(tgts):
        ctx.set_target(tc, ts, slot=s)
    n = int(rng.integers(3, 13))
    items, T0 = [], []
    for i in range(n):
        h, w = [(16, 300), (16, 900), (32, 900), (64, 450)][int(rng.integers(0, 4))]
        sc = synth.make_scan(h, w, 7000 + 50 * b + i, labelled=labelled)
        items.append(dict(src_corner=sc["corner"], src_surf=sc["surf"], target=int(rng.integers(0, 2))))
        T0.append(synth.perturb_pose(sc["T_true"], np.random.default_rng(b * 100 + i)))
    T0 = np.array(T0, np.float32)
    p = lisreg.default_params(variant)
    if rng.integers(0, 2): p.fixed_iters = int(rng.integers(2, 9))
    Tb, sb = ctx.align_batch(items, T0, p)
    fronts[ctx.get_option("front_end")] = fronts.get(ctx.get_option("front_end"), 0) + 1
    ok = True
    for i, it in enumerate(items):
        Ts, ss = ctx.align_batch([it], T0[i:i + 1], p)                # a batch of one: eight lanes per query, same target slot
        ok = ok and np.array_equal(Ts[0], Tb[i]) and ss[0] == sb[i]
    regs += n
    if not ok:
        bad.append(b); print(f"batch {b}: {n} registrations, variant {variant}: a single call DIFFERS from its place in the batch")
print(f"== {n_b - len(bad)} of {n_b} random batches ({regs} registrations; batch front-ends used {fronts}): every registration bit-identical to the same "
      f"registration run alone; differing batches: {bad}")
ctx.close()
