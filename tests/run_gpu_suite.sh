#!/bin/bash
# the driver's command, exit status printed on its own line (never behind a pipe or another command)
mkdir -p gpurun_out
python3 -m pytest tests/ -x -q -m gpu -p no:cacheprovider > gpurun_out/gpu_suite.log 2>&1
rc=$?
tail -15 gpurun_out/gpu_suite.log
echo "GPU_SUITE_RC=$rc"
python3 -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1
echo "SMOKE_RC=$?"
tail -3 gpurun_out/smoke.log
