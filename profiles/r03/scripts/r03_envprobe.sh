#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
for v in 0 1; do echo "HIP_FORCE_DEV_KERNARG=$v"; HIP_FORCE_DEV_KERNARG=$v python scripts/batch_probe.py; done
echo "unset"; python scripts/batch_probe.py
