#!/bin/bash
# evidence runs on the final tree (GPU box): tests/r06_sweeps.sh -> gpurun_out/r06sweeps/*.txt (last lines copied into profiles/)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=gpurun_out/r06sweeps; mkdir -p $O
timeout 900 python tests/exact_sweep.py 400 1000 search_mode=5 2>/dev/null | tail -1 > $O/r06_exact_sweep_cells_1000.txt
# the cell rows REBUILT inside every run: pair kernels, octant marks in LDS, rows filtered by the query marks — against the oracle to the bit
timeout 900 python tests/exact_sweep.py 100 500 search_mode=5,rebuild_targets_each_run=1 2>/dev/null | tail -1 > $O/r06_exact_sweep_cells_rebuilt_500.txt
timeout 600 python tests/exact_sweep.py 100 300 2>/dev/null | tail -1 > $O/r06_exact_sweep_walk_300.txt
timeout 600 python tests/exact_sweep.py 2000 300 variant=3 2>/dev/null | tail -1 > $O/r06_exact_sweep_v3_300.txt
timeout 900 python tests/frontend_sweep.py 0 1000 2>/dev/null | tail -2 > $O/r06_frontend_sweep_1000.txt
timeout 600 python tests/batch_sweep.py 300 2>/dev/null | tail -2 > $O/r06_batch_sweep_300.txt
timeout 900 python tests/chain_sweep.py 40 2>/dev/null | tail -6 > $O/r06_chain_sweep.txt
timeout 600 python tests/feature_sweep.py 300 2>/dev/null | tail -1 > $O/r06_feature_sweep_300.txt
timeout 600 python tests/icp_batch_sweep.py 2>/dev/null | tail -4 > $O/r06_icp_batch_sweep.txt
timeout 600 python tests/nonfinite_fuzz.py 2>/dev/null | tail -2 > $O/r06_nonfinite_fuzz.txt
timeout 600 python tests/nextrow_sweep.py 500 2>/dev/null | tail -1 > $O/r06_nextrow_sweep_500.txt
for f in $O/*.txt; do echo "== $f"; cat $f | cut -c1-400; done
