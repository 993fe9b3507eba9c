#!/bin/bash
# regenerates the bench lines / kernel tables kept under profiles/ (on the GPU box): tests/final_profiles.sh <tag>
TAG=${1:-r03}; R=$(pwd); O=$R/gpurun_out/final_$TAG; rm -rf $O; mkdir -p $O
for w in cfg1 cfg3 odom cfg4 cfg5; do timeout 300 python bench.py --workload $w 2>/dev/null | tail -1 > $O/${TAG}_bench_$w.json; done
timeout 300 python bench.py 2>/dev/null | tail -1 > $O/${TAG}_bench.json
timeout 120 python tests/replay_probe.py 2>/dev/null > $O/${TAG}_replay_stage_times.txt
( timeout 120 tests/ktrace.sh cfg3 30; timeout 120 tests/ktrace.sh odom 30 ) 2>&1 | grep -v "^W2026" > $O/${TAG}_replay_kernel_times.txt
timeout 900 bash tests/prof.sh $TAG > $O/prof.log 2>&1
cp gpurun_out/prof_$TAG/summary.txt $O/${TAG}_rocprofv3_summary.txt 2>/dev/null
ls $O
