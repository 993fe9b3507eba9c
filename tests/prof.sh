#!/bin/bash
# rocprofv3 recipe used for profiles/: kernel-trace stats pass + separate PMC passes (never combined with tracing).
# usage (on the GPU box, repo root): tests/prof.sh <tag> [bench args]     e.g.  tests/prof.sh r02
set -u
TAG=${1:-r02}; shift || true
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
export LISREG_BENCH_NO_EXACT=1 LISREG_BENCH_NO_OVERLAP=1      # the exact-arithmetic leg and the two-contexts leg of the bench line would add their own kernel rows to the tables
ARGS="--steps 3 --warmup 1 --cpu-regs 0 --no-profile --no-pcie --min-seconds 0 $*"
cd /tmp
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/trace -o trace -- python $REPO/bench.py $ARGS > $OUT/trace.log 2>&1
rocprofv3 --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU -d $OUT/pmc_sq -o pmc -- python $REPO/bench.py $ARGS > $OUT/pmc_sq.log 2>&1
rocprofv3 --output-format csv --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_THREAD_CYCLES_VALU -d $OUT/pmc_sq2 -o pmc -- python $REPO/bench.py $ARGS > $OUT/pmc_sq2.log 2>&1
rocprofv3 --output-format csv --pmc FETCH_SIZE -d $OUT/pmc_fetch -o pmc -- python $REPO/bench.py $ARGS > $OUT/pmc_fetch.log 2>&1
rocprofv3 --output-format csv --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d $OUT/pmc_write -o pmc -- python $REPO/bench.py $ARGS > $OUT/pmc_write.log 2>&1
rocprofv3 --output-format csv --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum -d $OUT/pmc_tcp -o pmc -- python $REPO/bench.py $ARGS > $OUT/pmc_tcp.log 2>&1
# the "next" rows of SURVEY.md §8 f (voxel grid, feature extraction, map filters, ICP): kernel-trace of the measurement probes
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/next_vox -o trace -- python $REPO/tests/voxel_probe.py > $OUT/next_vox.log 2>&1
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/next_feat -o trace -- python $REPO/tests/feature_probe.py > $OUT/next_feat.log 2>&1
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/next_map -o trace -- python $REPO/tests/mapfilter_probe.py > $OUT/next_map.log 2>&1
cd $REPO
python tests/prof_summarize.py $OUT $TAG > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
# keep only small files
find $OUT -size +2M -delete
