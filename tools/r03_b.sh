#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_rns.py tests/test_gpu_fuzz.py tests/test_gpu_workloads.py -x -q -m gpu > $OUT/r03b_pytest.txt 2>&1
tail -5 $OUT/r03b_pytest.txt
TAG=r03b_ks bash tools/ks_trace.sh
cat $OUT/ks_trace.log | tail -12
