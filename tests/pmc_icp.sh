#!/bin/bash
# experiment helper (GPU box): per-launch durations and per-wave instruction counters of k_icp_assoc (bench.py --workload cfg4_icp)
R=${GRAFT_REPO_ROOT:-$(pwd)}; export TMPDIR=/tmp; cd /tmp; rm -rf /tmp/icp_t /tmp/icp_p
rocprofv3 --output-format csv --kernel-trace -d /tmp/icp_t -o t -- python $R/bench.py --workload cfg4_icp --steps 2 --warmup 1 --cpu-regs 0 > /tmp/icp_t.log 2>&1
rocprofv3 --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVES SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES -d /tmp/icp_p -o pmc -- python $R/bench.py --workload cfg4_icp --steps 2 --warmup 1 --cpu-regs 0 > /tmp/icp_p.log 2>&1
python - <<'PY'
import csv,glob,collections
f=glob.glob('/tmp/icp_t/**/*kernel_trace.csv',recursive=True)[0]
rows=sorted([r for r in csv.DictReader(open(f)) if 'k_icp_assoc' in r['Kernel_Name']], key=lambda r:int(r['Start_Timestamp']))
d=[round((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3) for r in rows]
print('k_icp_assoc launches (us), in order:', d)
rows=collections.defaultdict(dict)
for fn in glob.glob('/tmp/icp_p/**/*counter_collection.csv',recursive=True):
    for r in csv.DictReader(open(fn)):
        if 'k_icp_assoc' in r['Kernel_Name']: rows[int(r['Dispatch_Id'])][r['Counter_Name']]=float(r['Counter_Value'])
for i in sorted(rows)[-22:]:
    m=rows[i]; w=max(m.get('SQ_WAVES',1),1)
    print(i,'waves',int(w),'VALU/wave',round(m['SQ_INSTS_VALU']/w),'SALU',round(m['SQ_INSTS_SALU']/w),'VMEM_RD',round(m['SQ_INSTS_VMEM_RD']/w,1),'LDS',round(m['SQ_INSTS_LDS']/w,1),'lane use',round(m['SQ_THREAD_CYCLES_VALU']/max(m['SQ_ACTIVE_INST_VALU'],1)/64,3))
PY
