for t in 8 12 15; do
  LISREG_BENCH_NO_EXACT=1 LISREG_OPTS=feeder_threads=$t python bench.py --steps 20 --warmup 5 --cpu-regs 0 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); p=d['pcie_inclusive']; print('threads $t', 'value', d['value'], 'pcie', p['value'], p['runs'], p.get('stage_ms'), p.get('chunks_taken_by_copy_engine'))"
done
