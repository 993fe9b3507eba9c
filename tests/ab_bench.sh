#!/bin/bash
# experiment helper (GPU box): tests/ab_bench.sh "label:VAR=val ..." ... — bench.py under each environment: value, per-step split (HIP events), walking wavefronts
R=${GRAFT_REPO_ROOT:-$(pwd)}
for spec in "$@"; do
  v=${spec%%:*}; envs=${spec#*:}
  env $envs LISREG_COUNT=1 python $R/bench.py --steps 10 --warmup 2 --cpu-regs 0 --no-pcie ${BENCH_ARGS:-} > /tmp/abb_$v.log 2> /tmp/abb_$v.err
  grep -h "wavefronts with" /tmp/abb_$v.err | cut -c1-200
  python - "$v" /tmp/abb_$v.log <<'PY'
import sys, json
v, f = sys.argv[1:3]
for l in open(f):
    if l.startswith("{"):
        d = json.loads(l); r = d.get("roofline") or {}
        print(f"{v:>12s} value {d['value']:.0f} ms/step {d['ms_per_step']} frac {r.get('frac')} avg_launch_ms {r.get('avg_launch_ms')} per_step {r.get('per_step_ms')}")
PY
done
