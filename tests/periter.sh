export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp
rocprofv3 --output-format csv --kernel-trace -d $R/gpurun_out/periter -o t -- python $R/bench.py --steps 3 --warmup 1 --cpu-regs 0 --no-profile > /dev/null 2>&1
cd $R
python - <<'PY'
import csv,glob,collections
f=glob.glob('gpurun_out/periter/**/*kernel_trace.csv',recursive=True)[0]
rows=[r for r in csv.DictReader(open(f)) if 'k_assoc_' in r['Kernel_Name']]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
d=collections.defaultdict(list)
for i,r in enumerate(rows): d[i%10].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
print('per-iteration us:', [round(sum(v)/len(v),1) for k,v in sorted(d.items())], 'mean', round(sum(sum(v) for v in d.values())/len(rows),1))
PY
rm -rf gpurun_out/periter
