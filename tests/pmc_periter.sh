#!/bin/bash
# per-GN-iteration instruction counts of k_assoc_walk (one rocprofv3 --pmc pass, no tracing): tests/pmc_periter.sh [bench args]
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/pmc_periter; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp LISREG_BENCH_NO_EXACT=1 LISREG_BENCH_NO_OVERLAP=1; cd /tmp
rocprofv3 --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVES SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_BUSY_CU_CYCLES -d $OUT -o pmc -- python $REPO/bench.py --steps 2 --warmup 1 --cpu-regs 0 --no-profile --no-pcie --min-seconds 0 "$@" > $OUT/log.txt 2>&1
cd $REPO
python - <<'PY'
import csv,glob,collections
rows=collections.defaultdict(dict)
for fn in glob.glob('gpurun_out/pmc_periter/**/*counter_collection.csv',recursive=True):
    for r in csv.DictReader(open(fn)):
        if 'k_assoc_' in r['Kernel_Name']:
            rows[int(r['Dispatch_Id'])][r['Counter_Name']]=float(r['Counter_Value'])
ids=sorted(rows)
per=collections.defaultdict(lambda: collections.defaultdict(list))
for n,i in enumerate(ids):
    for k,v in rows[i].items(): per[n%10][k].append(v)
print('iter  VALU/wave  SALU/wave  VMEM_RD/wave  LDS/wave  lane-use  ACTIVE_VALU/wave')
for it in range(10):
    m={k:sum(v)/len(v) for k,v in per[it].items()}
    w=max(m.get('SQ_WAVES',1),1)
    print(it, round(m['SQ_INSTS_VALU']/w,1), round(m['SQ_INSTS_SALU']/w,1), round(m['SQ_INSTS_VMEM_RD']/w,2), round(m['SQ_INSTS_LDS']/w,2),
          round(m['SQ_THREAD_CYCLES_VALU']/max(m['SQ_ACTIVE_INST_VALU'],1)/64,3), round(m['SQ_ACTIVE_INST_VALU']/w,1))
PY
find $OUT -size +2M -delete
