#!/bin/bash
# regenerates the bench lines / kernel tables kept under profiles/ (on the GPU box): tests/final_profiles.sh <tag>
TAG=${1:-r03}; R=$(pwd); O=$R/gpurun_out/final_$TAG; rm -rf $O; mkdir -p $O
for w in cfg1 cfg3 odom cfg4 cfg5 cfg4_icp; do timeout 300 python bench.py --workload $w 2>/dev/null | tail -1 > $O/${TAG}_bench_$w.json; done
timeout 300 python bench.py 2>/dev/null | tail -1 > $O/${TAG}_bench.json
LISREG_BENCH_FORCE_DIST=1 timeout 300 python bench.py --steps 10 --warmup 2 --cpu-regs 0 --no-pcie 2>/dev/null | tail -1 > $O/${TAG}_bench_force_dist_1rank.json
( cd /tmp; export TMPDIR=/tmp; rocprofv3 --output-format csv --kernel-trace --stats -d $O/icp_trace -o trace -- python $R/bench.py --workload cfg4_icp --steps 2 --warmup 1 --cpu-regs 0 > $O/icp_trace.log 2>&1 )
python - "$O" "$TAG" <<'PY'
import csv, glob, sys
O, TAG = sys.argv[1], sys.argv[2]
f = glob.glob(f"{O}/icp_trace/**/*kernel_stats.csv", recursive=True)
if f:
    rows = [r for r in csv.DictReader(open(f[0])) if "lisreg" in r["Name"]]
    with open(f"{O}/{TAG}_cfg4_icp_kernel_stats.csv", "w") as o:
        w = csv.DictWriter(o, fieldnames=list(rows[0].keys())); w.writeheader(); w.writerows(rows)
PY
rm -rf $O/icp_trace
timeout 120 python tests/replay_probe.py 2>/dev/null > $O/${TAG}_replay_stage_times.txt
( timeout 120 tests/ktrace.sh cfg3 30; timeout 120 tests/ktrace.sh odom 30 ) 2>&1 | grep -v "^W2026" > $O/${TAG}_replay_kernel_times.txt
timeout 900 bash tests/prof.sh $TAG > $O/prof.log 2>&1
cp gpurun_out/prof_$TAG/summary.txt $O/${TAG}_rocprofv3_summary.txt 2>/dev/null
ls $O
