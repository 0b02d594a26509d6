#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
TAG=${TAG:-r05d}
cd $R
T0=$(date +%s)
timeout 1500 python -m pytest tests -x -q -m gpu > $OUT/${TAG}_pytest.txt 2>&1
tail -4 $OUT/${TAG}_pytest.txt
echo "pytest seconds: $(( $(date +%s) - T0 ))"
SKIP_PYTEST=1 VARIANTS="mcsoff" bash tools/exp_mcs.sh
