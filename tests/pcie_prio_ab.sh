#!/bin/bash
# experiment helper (GPU box): PCIe-inclusive leg with / without the two legs in front of it (they leave streams behind that alias hardware queues),
# round 6's feeder against LISREG_FEED_LEGACY=1 (rounds 3-5: device-side wait + packing kernels on the copy stream), copy-stream priority as given
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
pr() { python -c "
import sys,json; d=json.loads(sys.stdin.read()); p=d['pcie_inclusive']; print('$1', 'value', d['value'], 'pcie', p['value'], p['runs'], 'stage', p.get('stage_ms'), 'series', p['in_series']['value'], p.get('chunks_taken_by_copy_engine','')[:12])"; }
for rep in 1 2; do
LISREG_BENCH_NO_EXACT=1 python bench.py --steps 20 --warmup 5 --cpu-regs 0 2>/dev/null | tail -1 | pr "default          "
LISREG_BENCH_NO_EXACT=1 LISREG_BENCH_NO_OVERLAP=1 python bench.py --steps 20 --warmup 5 --cpu-regs 0 2>/dev/null | tail -1 | pr "no_overlap       "
LISREG_FEED_LEGACY=1 LISREG_BENCH_NO_EXACT=1 python bench.py --steps 20 --warmup 5 --cpu-regs 0 2>/dev/null | tail -1 | pr "default legacy   "
LISREG_FEED_LEGACY=1 LISREG_BENCH_NO_EXACT=1 LISREG_BENCH_NO_OVERLAP=1 python bench.py --steps 20 --warmup 5 --cpu-regs 0 2>/dev/null | tail -1 | pr "no_overlap legacy"
done
