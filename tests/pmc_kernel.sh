#!/bin/bash
# experiment helper (GPU box): tests/pmc_kernel.sh <kernel-name-substring> "VAR=val ..." — per-wave instruction counters of one kernel (rocprofv3 --pmc, no tracing)
set -u
K=$1; envs=$2; REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=/tmp/pmc_kernel; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp LISREG_BENCH_NO_EXACT=1 LISREG_BENCH_NO_OVERLAP=1; cd /tmp
env $envs rocprofv3 --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVES SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES -d $OUT -o pmc -- python $REPO/bench.py --steps 2 --warmup 1 --cpu-regs 0 --no-profile --no-pcie --min-seconds 0 > $OUT/log.txt 2>&1
python - "$K" <<'PY'
import csv,glob,collections,sys
K=sys.argv[1]
rows=collections.defaultdict(dict)
for fn in glob.glob('/tmp/pmc_kernel/**/*counter_collection.csv',recursive=True):
    for r in csv.DictReader(open(fn)):
        if K in r['Kernel_Name']:
            rows[int(r['Dispatch_Id'])][r['Counter_Name']]=float(r['Counter_Value'])
for i in sorted(rows)[:6]:
    m=rows[i]; w=max(m.get('SQ_WAVES',1),1)
    print(i, 'waves', int(w), 'VALU/wave', round(m['SQ_INSTS_VALU']/w,1), 'SALU/wave', round(m['SQ_INSTS_SALU']/w,1), 'VMEM_RD/wave', round(m['SQ_INSTS_VMEM_RD']/w,2),
          'LDS/wave', round(m['SQ_INSTS_LDS']/w,2), 'lane-use', round(m['SQ_THREAD_CYCLES_VALU']/max(m['SQ_ACTIVE_INST_VALU'],1)/64,3), 'wave_cycles/wave', round(m.get('SQ_WAVE_CYCLES',0)/w,1))
PY
