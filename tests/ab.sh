#!/bin/bash
# experiment helper: tests/ab.sh <variant> [<variant> ...]  (on the GPU box) — per-GN-iteration k_assoc durations of each variant library
# (rocprofv3 --kernel-trace of bench.py), interleaved twice so that box-to-box and warm-up differences show.
R=${GRAFT_REPO_ROOT:-$(pwd)}; export TMPDIR=/tmp LISREG_BENCH_NO_OVERLAP=1 LISREG_BENCH_NO_EXACT=1
cp $R/lis-slam_amd/lib/liblisreg.so /tmp/liblisreg_keep.so
for rep in 1 2; do
for v in "$@"; do
  if [ "$v" = "base" ]; then cp /tmp/liblisreg_keep.so $R/lis-slam_amd/lib/liblisreg.so; else cp $R/lis-slam_amd/lib/variants/liblisreg_$v.so $R/lis-slam_amd/lib/liblisreg.so; fi
  rm -rf /tmp/periter_$v; cd /tmp
  rocprofv3 --output-format csv --kernel-trace -d /tmp/periter_$v -o t -- python $R/bench.py --steps 3 --warmup 1 --cpu-regs 0 --no-profile --no-pcie --min-seconds 0 ${BENCH_ARGS:-} > /tmp/periter_$v.log 2>&1
  cd $R
  python - "$v" <<'PY'
import csv,glob,collections,sys
v=sys.argv[1]
f=glob.glob(f'/tmp/periter_{v}/**/*kernel_trace.csv',recursive=True)[0]
allr=[r for r in csv.DictReader(open(f))]
rows=[r for r in allr if 'k_assoc_' in r['Kernel_Name']]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
n=int(__import__('os').environ.get('AB_ITERS','10'))
d=collections.defaultdict(list)
for i,r in enumerate(rows): d[i%n].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
oth=collections.defaultdict(float)
for r in allr:
    if 'k_assoc_' not in r['Kernel_Name']: oth[r['Kernel_Name'].split('(')[0][-40:]]+=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3
steps=max(len(rows)//n,1)
print(f'{v:>10s} per-iteration us:', [round(sum(x)/len(x),1) for k,x in sorted(d.items())], 'mean', round(sum(sum(x) for x in d.values())/max(len(rows),1),1),
      '| other kernels us/step:', {k:round(t/steps,1) for k,t in sorted(oth.items(), key=lambda kv:-kv[1])[:4]})
PY
done; done
cp /tmp/liblisreg_keep.so $R/lis-slam_amd/lib/liblisreg.so
