#!/bin/bash
# experiment helper (GPU box): tests/pmc_variants.sh "<ENV>" <variant> ... — per-wave instruction counters of k_assoc_walk for each variant
# library (base = the regular build), averaged over the launches of the timed steps (rocprofv3 --pmc, no tracing).
R=${GRAFT_REPO_ROOT:-$(pwd)}; envs=$1; shift
cp $R/lis-slam_amd/lib/liblisreg.so /tmp/liblisreg_keep.so
for v in "$@"; do
  if [ "$v" = "base" ]; then cp /tmp/liblisreg_keep.so $R/lis-slam_amd/lib/liblisreg.so; else cp $R/lis-slam_amd/lib/variants/liblisreg_$v.so $R/lis-slam_amd/lib/liblisreg.so; fi
  OUT=/tmp/pmc_v_$v; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp LISREG_BENCH_NO_EXACT=1 LISREG_BENCH_NO_OVERLAP=1
  ( cd /tmp; env $envs rocprofv3 --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVES SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU -d $OUT -o pmc -- python $R/bench.py --steps 2 --warmup 1 --cpu-regs 0 --no-profile --no-pcie --min-seconds 0 ${BENCH_ARGS:-} > $OUT/log.txt 2>&1 )
  python - "$v" <<'PY'
import csv,glob,collections,sys
v=sys.argv[1]
rows=collections.defaultdict(dict)
for fn in glob.glob(f'/tmp/pmc_v_{v}/**/*counter_collection.csv',recursive=True):
    for r in csv.DictReader(open(fn)):
        if 'k_assoc_walk' in r['Kernel_Name']:
            rows[int(r['Dispatch_Id'])][r['Counter_Name']]=float(r['Counter_Value'])
ids=sorted(rows)
per=[]
for i in ids:
    m=rows[i]; w=max(m.get('SQ_WAVES',1),1)
    per.append((m['SQ_INSTS_VALU']/w, m['SQ_INSTS_SALU']/w, m['SQ_INSTS_VMEM_RD']/w, m['SQ_THREAD_CYCLES_VALU']/max(m['SQ_ACTIVE_INST_VALU'],1)/64))
n=len(per)//3 if len(per)>=3 else 1          # launches per step (3 steps: 1 warm-up + 2 timed)
last=per[-n:]
print(f'{v:>12s} VALU/wave by launch of the last step:', [round(p[0]) for p in last], '| SALU', round(sum(p[1] for p in last)/len(last)), 'VMEM_RD', round(sum(p[2] for p in last)/len(last),1), 'lane use', round(sum(p[3] for p in last)/len(last),3))
PY
done
cp /tmp/liblisreg_keep.so $R/lis-slam_amd/lib/liblisreg.so
