#!/bin/bash
# round 5: the whole GPU test suite, the smoke entry, then the evidence pass (scripts/r05_profile.sh)
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
mkdir -p gpurun_out/r05
timeout 3000 python -m pytest tests -m gpu -q -x 2>&1 | tail -6 > gpurun_out/r05/pytest_gpu.txt; cat gpurun_out/r05/pytest_gpu.txt
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
bash scripts/r05_profile.sh > gpurun_out/r05_profile.log 2>&1; tail -12 gpurun_out/r05_profile.log | cut -c1-300
