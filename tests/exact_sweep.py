"""Evidence run (not a test): the exact-arithmetic build against the oracle on N random configurations of the sweep generator
(tests/test_exact.py::sweep_case with seeds beyond the 12 of the test suite).  Prints one line per seed and a summary.
  python tests/exact_sweep.py [first_seed] [n]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "lis-slam_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import lisreg
import oracle_ctypes as oc
from test_exact import sweep_case, check_exact
oc.build()
first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
n = int(sys.argv[2]) if len(sys.argv) > 2 else 100
# optional: name=value,... options for the library context (search_mode=3, lanes_per_query=1) and variant=3 to force copy #3's parameters
extra = dict(kv.split("=") for kv in (sys.argv[3].split(",") if len(sys.argv) > 3 else []))
force_variant = int(extra.pop("variant", 0))
opts = tuple((k, int(v)) for k, v in extra.items())
bit_identical = 0; flags = 0; fails = []
for seed in range(first, first + n):
    case, variant, fixed, imu = sweep_case(seed)
    if force_variant:
        variant = force_variant
        if variant == 3: imu = None
    p_o = oc.default_params(variant); p_o.fixed_iters = fixed
    try:
        worst, k = check_exact(oc, lisreg, case, p_o, imu, opts=opts)
        bit_identical += worst == 0.0; flags += k
        print(f"seed {seed}: variant {variant} fixed_iters {fixed} imu {imu is not None}: integer outputs equal, worst pose difference {worst:.2e}, {k} accept flags compared")
    except AssertionError as e:
        fails.append(seed); print(f"seed {seed}: MISMATCH {str(e)[:300]}")
print(f"== options {dict(opts)} variant {force_variant or 'as drawn'}: {n - len(fails)} of {n} configurations: status / isDegenerate / iteration count / n_corr of every iteration / accept flags equal to the oracle's; "
      f"poses bit-identical in {bit_identical}; {flags} accept flags compared; mismatching seeds: {fails}")
