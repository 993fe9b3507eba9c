#!/bin/bash
# experiment helper (GPU box): tests/timeline.sh [bench args] — the kernels of the LAST step of a short bench run as a timeline
# (start offset, duration, queue) from rocprofv3 --kernel-trace
R=${GRAFT_REPO_ROOT:-$(pwd)}; export TMPDIR=/tmp LISREG_BENCH_NO_EXACT=1 LISREG_BENCH_NO_OVERLAP=1; cd /tmp; rm -rf /tmp/tl
rocprofv3 --output-format csv --kernel-trace -d /tmp/tl -o t -- python $R/bench.py --steps 3 --warmup 1 --cpu-regs 0 --no-profile --no-pcie --min-seconds 0 "$@" > /tmp/tl.log 2>&1
python - <<'PY'
import csv,glob,re
f=glob.glob('/tmp/tl/**/*kernel_trace.csv',recursive=True)[0]
rows=sorted([r for r in csv.DictReader(open(f)) if 'lisreg' in r['Kernel_Name'] or 'rocclr' in r['Kernel_Name']], key=lambda r:int(r['Start_Timestamp']))
# last step = from the last k_strip_partition<false> (or k_reset_items) on
idx=[i for i,r in enumerate(rows) if 'k_strip_partition' in r['Kernel_Name']]
start=idx[-2] if len(idx)>=2 else 0
t0=int(rows[start]['Start_Timestamp'])
for r in rows[start:]:
    m=re.search(r'(k_\w+(<[^>(]*>)?)',r['Kernel_Name'])
    s=(int(r['Start_Timestamp'])-t0)/1e3; d=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3
    print(f"{s:9.1f} us  +{d:7.1f}  q{r.get('Queue_Id','?'):>3s}  {m.group(1) if m else r['Kernel_Name'][:40]}")
PY
