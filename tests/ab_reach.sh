#!/bin/bash
# experiment helper (GPU box): option row_reach on / off — bench value without the profiler (3 interleaved pairs), walking wavefronts per GN iteration, timeline
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; export LISREG_BENCH_NO_EXACT=1 LISREG_BENCH_NO_OVERLAP=1
for rep in 1 2 3; do for r in 0 1; do
  LISREG_ROW_REACH=$r python bench.py --steps 20 --warmup 5 --cpu-regs 0 --no-pcie 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('row_reach $r value', d['value'], 'ms/step', d['ms_per_step'], 'frac', d['roofline']['frac'], 'step_frac', d['roofline'].get('step_frac'))"
done; done
for r in 0 1; do echo "row_reach $r:"; LISREG_ROW_REACH=$r LISREG_COUNT=1 python bench.py --steps 3 --warmup 1 --cpu-regs 0 --no-pcie --no-profile --min-seconds 0 2>&1 >/dev/null | grep -i "walking\|searched"; done
LISREG_ROW_REACH=1 bash tests/timeline.sh
