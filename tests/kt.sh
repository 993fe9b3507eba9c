#!/bin/bash
# per-kernel times of one bench workload (rocprofv3 kernel trace): tests/kt.sh [bench args]
export TMPDIR=/tmp; R=$(pwd); cd /tmp; rm -rf $R/gpurun_out/kt
rocprofv3 --output-format csv --kernel-trace --stats -d $R/gpurun_out/kt -o t -- python $R/bench.py --steps 3 --warmup 1 --cpu-regs 0 --no-profile --no-pcie --min-seconds 0 "$@" > /dev/null 2>&1
cd $R
python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/kt/**/*kernel_stats.csv',recursive=True)[0]
for r in [r for r in csv.DictReader(open(f)) if 'lisreg' in r['Name']][:24]:
    print(f"{r['Name'].replace('void lisreg::(anonymous namespace)::','')[:70]:70s} calls {r['Calls']:>6s} avg_us {float(r['AverageNs'])/1e3:9.1f} total_ms {float(r['TotalDurationNs'])/1e6:8.2f} {r['Percentage']}%")
PY
rm -rf gpurun_out/kt
