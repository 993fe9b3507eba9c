#!/bin/bash
# experiment helper (GPU box): tests/kstat.sh "VAR=val ..." [bench args] — rocprofv3 kernel stats (mean us per kernel) of bench.py --steps 3
R=${GRAFT_REPO_ROOT:-$(pwd)}; export TMPDIR=/tmp; export LISREG_BENCH_NO_EXACT=1 LISREG_BENCH_NO_OVERLAP=1; envs=$1; shift
rm -rf /tmp/kstat; cd /tmp
env $envs rocprofv3 --output-format csv --kernel-trace --stats -d /tmp/kstat -o t -- python $R/bench.py --steps 3 --warmup 1 --cpu-regs 0 --no-profile --no-pcie --min-seconds 0 "$@" > /tmp/kstat.log 2>&1
python - <<'PY'
import csv,glob
f=glob.glob('/tmp/kstat/**/*kernel_stats.csv',recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'lisreg' in r['Name'] or 'k_' in r['Name']:
        print(f"{r['Name'][:90]:90s} calls {r['Calls']:>5s} avg_us {float(r['AverageNs'])/1e3:9.1f} total_ms {float(r['TotalDurationNs'])/1e6:8.3f}")
PY
