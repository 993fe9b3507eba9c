#!/bin/bash
# experiment helper (GPU box): tests/ab_breakeven.sh — graph scan (3) against cell rows (5) by batch size on the configs[1] scene (where should auto switch?)
export LISREG_BENCH_NO_EXACT=1 LISREG_BENCH_NO_OVERLAP=1
for b in ${BATCHES:-16 20 24 32 40}; do for m in 3 5; do
  LISREG_SEARCH_MODE=$m python bench.py --batch $b --cpu-regs 0 --no-pcie ${BENCH_ARGS:-} 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('batch $b mode $m', d['value'], d['ms_per_step'], d['roofline'].get('per_step_ms'))"
done; done
