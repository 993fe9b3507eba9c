set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/pmc_lanes; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
rocprofv3 --output-format csv --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $OUT -o pmc -- python $REPO/bench.py --steps 2 --warmup 1 --cpu-regs 0 --no-profile > $OUT/log.txt 2>&1
cd $REPO
python - <<'PY'
import csv,glob,collections
f=glob.glob('gpurun_out/pmc_lanes/**/*counter_collection.csv',recursive=True)
print(f)
acc=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for fn in f:
    for r in csv.DictReader(open(fn)):
        k=r['Kernel_Name'][:40]; acc[k][r['Counter_Name']]+=float(r['Counter_Value'])
for k,v in acc.items():
    if 'assoc' in k or 'solve' in k:
        print(k, {a: round(b) for a,b in v.items()})
        if v.get('SQ_ACTIVE_INST_VALU'): print('  lane util', v['SQ_THREAD_CYCLES_VALU']/(v['SQ_ACTIVE_INST_VALU']*64), ' valu/wave', v['SQ_INSTS_VALU']/max(v['SQ_WAVES'],1), 'salu/wave', v['SQ_INSTS_SALU']/max(v['SQ_WAVES'],1))
PY
find $OUT -size +2M -delete
