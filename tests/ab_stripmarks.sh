#!/bin/bash
# (the option it switches exists only with profiles/r06_xp_strip_marks.patch applied: the variant was slower and is not in the tree)
# experiment helper (GPU box): octant marks inside the strip builds (default) against a marks launch of their own (LISREG_STRIP_MARKS=0)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; export LISREG_BENCH_NO_EXACT=1 LISREG_BENCH_NO_OVERLAP=1
for rep in 1 2 3; do for r in 0 1; do
  LISREG_STRIP_MARKS=$r python bench.py --steps 20 --warmup 5 --cpu-regs 0 --no-pcie 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('strip_marks $r value', d['value'], 'ms/step', d['ms_per_step'], 'step_frac', d['roofline'].get('step_frac'))"
done; done
bash tests/timeline.sh | head -12
