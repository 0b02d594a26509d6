#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests/test_gpu_workloads.py tests/test_gpu_rns.py tests/test_gpu_fuzz.py tests/test_gpu_pyphantom.py tests/test_gpu_host_api.py -x -q -m gpu > $OUT/r03f_pytest.txt 2>&1
tail -4 $OUT/r03f_pytest.txt
timeout 300 python tools/time_hoist.py 2>&1 | grep hoisting
timeout 300 python tools/time_matvec.py 2>&1 | grep "hoisted + weighted"
