#!/bin/bash
# experiment helper (GPU box): tests/stepgap.sh — what runs between the k_finalize of one configs[1] step and the first kernel of the next (rocprofv3 --kernel-trace)
R=${GRAFT_REPO_ROOT:-$(pwd)}; export TMPDIR=/tmp LISREG_BENCH_NO_EXACT=1 LISREG_BENCH_NO_OVERLAP=1; cd /tmp; rm -rf /tmp/sg
rocprofv3 --output-format csv --kernel-trace --memory-copy-trace -d /tmp/sg -o t -- python $R/bench.py --steps 6 --warmup 1 --cpu-regs 0 --no-profile --no-pcie --min-seconds 0 "$@" > /tmp/sg.log 2>&1
python - <<'PY'
import csv,glob,re
f=glob.glob('/tmp/sg/**/*kernel_trace.csv',recursive=True)[0]
rows=sorted(csv.DictReader(open(f)), key=lambda r:int(r['Start_Timestamp']))
fin=[i for i,r in enumerate(rows) if 'k_finalize' in r['Kernel_Name']]
for i in fin[-5:-1]:
    e=int(rows[i]['End_Timestamp'])
    print('after k_finalize:', [(re.sub(r'\(.*','',rows[j]['Kernel_Name'].split('::')[-1])[:28], round((int(rows[j]['Start_Timestamp'])-e)/1e3,1), round((int(rows[j]['End_Timestamp'])-int(rows[j]['Start_Timestamp']))/1e3,1)) for j in range(i+1,min(i+4,len(rows)))])
s=[int(r['Start_Timestamp']) for r in rows if 'k_strip_partition<false>' in r['Kernel_Name'] or 'k_strip_partitionILb0' in r['Kernel_Name']]
print('step period us:', [round((b-a)/1e3,1) for a,b in zip(s[-6:],s[-5:])])
PY
