#!/bin/bash
# per-GN-iteration PMC of the k_assoc launches, any counter set (one rocprofv3 --pmc pass, no tracing):
#   COUNTERS="SQ_WAVES SQ_WAVE_CYCLES ..." tests/pmc_iter.sh [bench args]        values are printed per wave of the dispatch
set -u
REPO=$(pwd); OUT=/tmp/pmc_iter; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
rocprofv3 --output-format csv --pmc ${COUNTERS:-SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU} -d $OUT -o pmc -- python $REPO/bench.py --steps 2 --warmup 1 --cpu-regs 0 --no-profile --no-pcie --min-seconds 0 "$@" > $OUT/log.txt 2>&1
cd $REPO
python - <<'PY'
import csv,glob,collections,os
rows=collections.defaultdict(dict)
for fn in glob.glob('/tmp/pmc_iter/**/*counter_collection.csv',recursive=True):
    for r in csv.DictReader(open(fn)):
        if 'k_assoc_' in r['Kernel_Name']:
            rows[int(r['Dispatch_Id'])][r['Counter_Name']]=float(r['Counter_Value'])
ids=sorted(rows)
n=int(os.environ.get('AB_ITERS','10'))
per=collections.defaultdict(lambda: collections.defaultdict(list))
for k,i in enumerate(ids):
    for c,v in rows[i].items(): per[k%n][c].append(v)
names=sorted({c for it in per.values() for c in it})
print('iter '+' '.join(f'{c[-22:]:>22s}' for c in names))
waves=115200
for it in range(n):
    m={c:sum(v)/len(v) for c,v in per[it].items()}
    w=max(m.get('SQ_WAVES',waves),1); waves=w
    print(f'{it:4d} '+' '.join(f'{(m[c]/w if c!="SQ_WAVES" else m[c]):22.2f}' for c in names))
PY
