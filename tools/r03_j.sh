#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests/test_gpu_rns.py tests/test_gpu_fuzz.py tests/test_gpu_workloads.py -x -q -m gpu > $OUT/r03j_pytest.txt 2>&1
grep -E "passed|failed|Error" $OUT/r03j_pytest.txt | tail -3
timeout 300 python tools/ckks_ops_bench.py 2>&1 | grep "|"
PHA_OPS_LOGN=15 timeout 300 python tools/ckks_ops_bench.py 2>&1 | grep "|"
TAG=r03j_ks bash tools/ks_trace.sh | grep -E "modup_ip|inner_prod|bconv_kernel|16x60x3"
