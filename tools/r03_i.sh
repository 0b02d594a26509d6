#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests/test_gpu_workloads.py tests/test_gpu_ntt.py -x -q -m gpu > $OUT/r03i_pytest.txt 2>&1
grep -E "passed|failed" $OUT/r03i_pytest.txt | tail -2
timeout 600 python tools/time_bsgs.py 16x8 32x4 64x2 2>&1 | grep -E "steps|same"
