#!/bin/bash
# experiment helper (GPU box): AB_PAT="crow|solve" tests/ab_kernels.sh <variant> ... — per-step time of the kernels whose names match AB_PAT
# (and of all lisreg kernels) for each variant library (base = the regular build), rocprofv3 --kernel-trace of bench.py, interleaved twice.
R=${GRAFT_REPO_ROOT:-$(pwd)}; export TMPDIR=/tmp; export LISREG_BENCH_NO_EXACT=1 LISREG_BENCH_NO_OVERLAP=1
cp $R/lis-slam_amd/lib/liblisreg.so /tmp/liblisreg_keep.so
for rep in 1 2; do
for v in "$@"; do
  if [ "$v" = "base" ]; then cp /tmp/liblisreg_keep.so $R/lis-slam_amd/lib/liblisreg.so; else cp $R/lis-slam_amd/lib/variants/liblisreg_$v.so $R/lis-slam_amd/lib/liblisreg.so; fi
  rm -rf /tmp/abk_$v; cd /tmp
  rocprofv3 --output-format csv --kernel-trace -d /tmp/abk_$v -o t -- python $R/bench.py --steps 4 --warmup 1 --cpu-regs 0 --no-profile --no-pcie --min-seconds 0 ${BENCH_ARGS:-} > /tmp/abk_$v.log 2>&1
  cd $R
  python - "$v" <<'PY'
import csv,glob,collections,sys,os,re,json
v=sys.argv[1]
f=glob.glob(f'/tmp/abk_{v}/**/*kernel_trace.csv',recursive=True)[0]
pat=re.compile(os.environ.get('AB_PAT','crow'))
tot=collections.defaultdict(float); cnt=collections.defaultdict(int)
for r in csv.DictReader(open(f)):
    n=r['Kernel_Name']
    if 'lisreg' not in n: continue
    short=re.sub(r'\(.*','',n.replace('(anonymous namespace)::','')).split('::')[-1][:48]
    d=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3
    tot[short]+=d; cnt[short]+=1
steps=5.0
sel={k:(round(t/steps,1), round(t/cnt[k],1), cnt[k]) for k,t in tot.items() if pat.search(k)}
print(f'{v:>10s} us/step (avg us, calls):', sel, '| all lisreg kernels us/step', round(sum(tot.values())/steps,1))
try:
    l=[x for x in open(f'/tmp/abk_{v}.log') if x.startswith('{')][-1]; d=json.loads(l); print(f'{"":>10s} value', d['value'], 'ms/step', d['ms_per_step'])
except Exception as e: pass
PY
done; done
cp /tmp/liblisreg_keep.so $R/lis-slam_amd/lib/liblisreg.so
