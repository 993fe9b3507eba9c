#!/bin/bash
# experiment helper (GPU box): tests/ab_lib.sh <variant> <variant> ... — bench.py value with each lis-slam_amd/lib/variants/liblisreg_<variant>.so, three interleaved rounds, no profiler
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; export LISREG_BENCH_NO_EXACT=1 LISREG_BENCH_NO_OVERLAP=1
cp lis-slam_amd/lib/liblisreg.so /tmp/liblisreg_keep.so
for rep in 1 2 3; do for v in "$@"; do
  cp lis-slam_amd/lib/variants/liblisreg_$v.so lis-slam_amd/lib/liblisreg.so
  python bench.py --steps 20 --warmup 5 --cpu-regs 0 --no-pcie ${BENCH_ARGS:-} 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$v value', d['value'], 'ms/step', d['ms_per_step'], 'step_frac', d['roofline'].get('step_frac'))"
done; done
cp /tmp/liblisreg_keep.so lis-slam_amd/lib/liblisreg.so
