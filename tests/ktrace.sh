#!/bin/bash
# tests/ktrace.sh <workload> <steps>: kernel totals per frame of a replay workload
W=$1; S=${2:-30}
R=${GRAFT_REPO_ROOT:-$(pwd)}; export TMPDIR=/tmp; cd /tmp; rm -rf /tmp/kt_$W
rocprofv3 --output-format csv --kernel-trace -d /tmp/kt_$W -o t -- python $R/bench.py --workload $W --steps $S --warmup 2 > /tmp/kt_$W.log 2>&1
tail -1 /tmp/kt_$W.log | cut -c1-200
python - $W $S <<'PY'
import csv,glob,collections,sys
w=sys.argv[1]; steps=2*(int(sys.argv[2])+2)      # the bench line walks the frames twice: the timed pass and the untimed pass with events around the correspondence launches
f=glob.glob(f'/tmp/kt_{w}/**/*kernel_trace.csv',recursive=True)[0]
d=collections.defaultdict(lambda:[0,0.0])
tot=0
for r in csv.DictReader(open(f)):
    import re
    m=re.search(r'(k_\w+(<[^>(]*>)?|__amd_\w+)', r['Kernel_Name']); k=(m.group(1) if m else r['Kernel_Name'])[:44]
    t=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3
    d[k][0]+=1; d[k][1]+=t; tot+=t
print(f'{w}: kernel time per frame {tot/steps:.1f} us, launches per frame {sum(v[0] for v in d.values())/steps:.1f}')
for k,v in sorted(d.items(), key=lambda kv:-kv[1][1])[:22]: print(f'  {k:46s} {v[0]/steps:6.1f} launches/frame {v[1]/steps:8.1f} us/frame')
PY
python - $W <<'PY'
import csv,glob,sys
w=sys.argv[1]
f=glob.glob(f'/tmp/kt_{w}/**/*kernel_trace.csv',recursive=True)[0]
rows=sorted(csv.DictReader(open(f)), key=lambda r:int(r['Start_Timestamp']))
seq=[(r['Kernel_Name'], (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3) for r in rows]
a=[round(t,1) for k,t in seq if 'k_assoc_walk' in k]
print('k_assoc_walk launch durations, last 24 (us):', a[-24:])
PY
