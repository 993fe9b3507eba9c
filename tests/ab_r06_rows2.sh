#!/bin/bash
export LISREG_BENCH_NO_EXACT=1 LISREG_BENCH_NO_OVERLAP=1
pr() { python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print('$1', d['value'], d['ms_per_step'], 'frac', r.get('frac'), 'step_frac', r.get('step_frac'), 'front_end', r.get('search_front_end'), r.get('row_reach'))"; }
python bench.py --cpu-regs 0 --no-pcie --steps 20 --warmup 5 2>/dev/null | pr "cfg2 default"
for b in 8 16 32; do for m in 3 5; do
  LISREG_SEARCH_MODE=$m python bench.py --workload cfg5 --batch $b --cpu-regs 0 --no-pcie --steps 8 --warmup 2 2>/dev/null | pr "cfg5 batch $b mode $m"
done; done
LISREG_SEARCH_MODE=5 LISREG_ROW_REACH=0 python bench.py --workload cfg5 --batch 8 --cpu-regs 0 --no-pcie --steps 8 --warmup 2 2>/dev/null | pr "cfg5 batch 8 mode 5 no reach"
python -m pytest tests/test_round6_edges.py tests/test_gpu_batch.py -m gpu -x -q 2>&1 | grep -E "passed|failed"
