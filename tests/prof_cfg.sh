#!/bin/bash
# rocprofv3 summaries of one bench workload (kernel-trace stats pass + separate PMC passes): tests/prof_cfg.sh <workload> [iters]
set -u
W=${1:-cfg5}; IT=${2:-30}
REPO=$(pwd); OUT=$REPO/gpurun_out/prof_$W; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp LISREG_BENCH_NO_EXACT=1 LISREG_BENCH_NO_OVERLAP=1
ARGS="--workload $W --steps 3 --warmup 1 --cpu-regs 0 --no-profile --no-pcie --min-seconds 0"
cd /tmp
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/trace -o trace -- python $REPO/bench.py $ARGS > $OUT/trace.log 2>&1
rocprofv3 --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD -d $OUT/pmc_sq -o pmc -- python $REPO/bench.py $ARGS > $OUT/pmc_sq.log 2>&1
rocprofv3 --output-format csv --pmc FETCH_SIZE -d $OUT/pmc_fetch -o pmc -- python $REPO/bench.py $ARGS > $OUT/pmc_fetch.log 2>&1
rocprofv3 --output-format csv --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d $OUT/pmc_write -o pmc -- python $REPO/bench.py $ARGS > $OUT/pmc_write.log 2>&1
cd $REPO
python tests/prof_summarize.py $OUT r03_$W > $OUT/summary.txt 2>&1
AB_ITERS=$IT BENCH_ARGS="--workload $W" bash tests/ab.sh base 2>/dev/null | tail -1 >> $OUT/summary.txt
head -12 $OUT/summary.txt; tail -1 $OUT/summary.txt | cut -c1-300
find $OUT -size +2M -delete
