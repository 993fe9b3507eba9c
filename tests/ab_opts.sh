#!/bin/bash
# experiment helper (GPU box): tests/ab_opts.sh "label:name=value,name=value" ... — bench.py under each LISREG_OPTS setting, per-kernel time of AB_PAT kernels, twice
R=${GRAFT_REPO_ROOT:-$(pwd)}; export TMPDIR=/tmp LISREG_BENCH_NO_EXACT=1 LISREG_BENCH_NO_OVERLAP=1
for rep in 1 2; do for spec in "$@"; do
  v=${spec%%:*}; o=${spec#*:}; rm -rf /tmp/abo_$v; cd /tmp
  LISREG_OPTS="$o" rocprofv3 --output-format csv --kernel-trace -d /tmp/abo_$v -o t -- python $R/bench.py --steps 4 --warmup 1 --cpu-regs 0 --no-profile --no-pcie --min-seconds 0 ${BENCH_ARGS:-} > /tmp/abo_$v.log 2>&1
  cd $R
  python - "$v" <<'PY'
import csv,glob,collections,sys,os,re,json
v=sys.argv[1]
f=glob.glob(f'/tmp/abo_{v}/**/*kernel_trace.csv',recursive=True)[0]
pat=re.compile(os.environ.get('AB_PAT','strip'))
tot=collections.defaultdict(float); cnt=collections.defaultdict(int)
for r in csv.DictReader(open(f)):
    n=r['Kernel_Name']
    if 'lisreg' not in n: continue
    short=re.sub(r'\(.*','',n.replace('(anonymous namespace)::','')).split('::')[-1][:40]
    d=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3; tot[short]+=d; cnt[short]+=1
print(f'{v:>10s}', {k:(round(t/5.0,1), cnt[k]) for k,t in tot.items() if pat.search(k)}, end=' ')
try:
    l=[x for x in open(f'/tmp/abo_{v}.log') if x.startswith('{')][-1]; d=json.loads(l); print('value', d['value'], 'ms/step', d['ms_per_step'])
except Exception as e: print()
PY
done; done
