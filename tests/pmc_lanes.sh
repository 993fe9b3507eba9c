#!/bin/bash
# VALU instruction count and lane utilisation of the lisreg kernels (one rocprofv3 --pmc pass, no tracing).
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/pmc_lanes; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
rocprofv3 --output-format csv --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES SQ_INSTS_SALU SQ_INSTS_LDS -d $OUT -o pmc -- python $REPO/bench.py --steps 2 --warmup 1 --cpu-regs 0 --no-profile > $OUT/log.txt 2>&1
cd $REPO
python - <<'PY'
import csv,glob,collections
acc=collections.defaultdict(lambda: collections.defaultdict(float))
for fn in glob.glob('gpurun_out/pmc_lanes/**/*counter_collection.csv',recursive=True):
    for r in csv.DictReader(open(fn)):
        acc[r['Kernel_Name'][:60]][r['Counter_Name']]+=float(r['Counter_Value'])
for k,v in acc.items():
    if 'lisreg' in k and v.get('SQ_ACTIVE_INST_VALU'):
        print(k[-40:], 'lane use %.3f' % (v['SQ_THREAD_CYCLES_VALU']/(v['SQ_ACTIVE_INST_VALU']*64)), 'VALU/wave %.0f' % (v['SQ_INSTS_VALU']/max(v['SQ_WAVES'],1)),
              'SALU/wave %.0f' % (v['SQ_INSTS_SALU']/max(v['SQ_WAVES'],1)), 'LDS/wave %.1f' % (v['SQ_INSTS_LDS']/max(v['SQ_WAVES'],1)), 'busy quad-cycles/wave %.0f' % (v['SQ_ACTIVE_INST_VALU']/max(v['SQ_WAVES'],1)))
PY
find $OUT -size +2M -delete
