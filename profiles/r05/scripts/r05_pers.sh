#!/bin/bash
# round 5: persistent bin blocks — parity (quick subset), then the block-count sweep at configs[3] for two builds
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r05_pers
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_parity_gpu.py -m gpu -x -q 2>&1 | tail -5 > $O/pytest.txt
cat $O/pytest.txt
for N in "" w5; do
  L=$R/fastdem_amd/lib/libfdm_engine${N:+_$N}.so
  echo "== ${N:-shipped}"
  FDM_ENGINE_LIB=$L timeout 600 python scripts/c4_ab.py "" "bin_blocks=768" "bin_blocks=1280" "bin_blocks=2048" "bin_blocks=768,upd_blocks=768" "bin_blocks=1024,upd_blocks=256" "overlap=0" 2>/dev/null | tail -1
done
