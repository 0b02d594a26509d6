#!/bin/bash
# r06: A/B of the word-wise (base 2^30) Montgomery reduction of the conversions' split accumulators (mont_redc90_split) against the
# recombination + 64-bit REDC it replaces: full GPU suite on the new build, then exp_mcs.sh's timing + per-kernel trace with
# VARIANTS / TRACE = "base new" (libphantom_amd_base.so = the previous commit's three sources, libphantom_amd_new.so = a copy of the product)
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
TAG=${TAG:-r06f}
mkdir -p $OUT
cd $R
T0=$(date +%s)
timeout 1500 python -m pytest tests -x -q -m gpu > $OUT/${TAG}_pytest.txt 2>&1
tail -3 $OUT/${TAG}_pytest.txt
echo "pytest seconds: $(( $(date +%s) - T0 ))"
SKIP_PYTEST=1 TAG=$TAG VARIANTS="base new" TRACE="base new" bash tools/exp_mcs.sh
