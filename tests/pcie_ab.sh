#!/bin/bash
# experiment helper (GPU box): PCIe-inclusive leg by number of packing threads (LISREG_OPTS reaches the leg's context)
for t in 8 12 15; do
  LISREG_BENCH_NO_EXACT=1 LISREG_BENCH_NO_OVERLAP=1 LISREG_OPTS=feeder_threads=$t python bench.py --steps 20 --warmup 5 --cpu-regs 0 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); p=d['pcie_inclusive']; print('threads $t', 'value', d['value'], 'pcie', p['value'], p['runs'], 'stage', p.get('stage_ms'), 'series', p['in_series']['value'], p['feeder'])"
done
