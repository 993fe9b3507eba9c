#!/bin/bash
# L2 hit rate of the k_assoc_ launches (one rocprofv3 --pmc pass, no tracing): tests/pmc_l2.sh [bench args]
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/pmc_l2; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
rocprofv3 --output-format csv --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum -d $OUT -o pmc -- python $REPO/bench.py --steps 2 --warmup 1 --cpu-regs 0 --no-profile --no-pcie --min-seconds 0 "$@" > $OUT/log.txt 2>&1
cd $REPO
python - <<'PY'
import csv,glob,collections
rows=collections.defaultdict(dict)
for fn in glob.glob('gpurun_out/pmc_l2/**/*counter_collection.csv',recursive=True):
    for r in csv.DictReader(open(fn)):
        if 'k_assoc_' in r['Kernel_Name']:
            rows[int(r['Dispatch_Id'])][r['Counter_Name']]=float(r['Counter_Value'])
ids=sorted(rows)
for n,i in enumerate(ids[:10]):
    m=rows[i]; print(n, {k:round(v/1e6,2) for k,v in m.items()}, 'hit', round(m['TCC_HIT_sum']/max(m['TCC_HIT_sum']+m['TCC_MISS_sum'],1),3))
PY
find $OUT -size +2M -delete
