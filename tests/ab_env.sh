#!/bin/bash
# experiment helper (GPU box): tests/ab_env.sh "label:VAR=val VAR2=val2" ... — per-GN-iteration k_assoc_* durations of bench.py under each
# environment (rocprofv3 --kernel-trace), interleaved twice.  AB_ITERS = GN iterations per step (10), BENCH_ARGS = extra bench.py arguments.
R=${GRAFT_REPO_ROOT:-$(pwd)}; export TMPDIR=/tmp; export LISREG_BENCH_NO_EXACT=1 LISREG_BENCH_NO_OVERLAP=1      # (the exact-arithmetic leg of the bench line would put its own launches into the table)
for rep in 1 2; do
for spec in "$@"; do
  v=${spec%%:*}; envs=${spec#*:}
  rm -rf /tmp/periter_$v; cd /tmp
  env $envs rocprofv3 --output-format csv --kernel-trace -d /tmp/periter_$v -o t -- python $R/bench.py --steps 3 --warmup 1 --cpu-regs 0 --no-profile --no-pcie --min-seconds 0 ${BENCH_ARGS:-} > /tmp/periter_$v.log 2>&1
  cd $R
  python - "$v" <<'PY'
import csv,glob,collections,sys,os
v=sys.argv[1]
f=glob.glob(f'/tmp/periter_{v}/**/*kernel_trace.csv',recursive=True)[0]
allr=[r for r in csv.DictReader(open(f))]
rows=[r for r in allr if 'k_assoc_' in r['Kernel_Name']]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
n=int(os.environ.get('AB_ITERS','10'))
d=collections.defaultdict(list)
for i,r in enumerate(rows): d[i%n].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
print(f'{v:>12s} per-iteration us:', [round(sum(x)/len(x),1) for k,x in sorted(d.items())], 'mean', round(sum(sum(x) for x in d.values())/max(len(rows),1),1))
PY
  grep -h '"value"' /tmp/periter_$v.log | python -c "import sys,json; [print('             value', json.loads(l)['value'], 'ms/step', json.loads(l)['ms_per_step']) for l in sys.stdin]" 2>/dev/null
done; done
