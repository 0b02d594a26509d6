#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests/test_gpu_ntt.py tests/test_gpu_rns.py tests/test_gpu_comm.py tests/test_gpu_pyphantom.py tests/test_gpu_host_api.py -x -q -m gpu > $OUT/r03g_pytest.txt 2>&1
tail -4 $OUT/r03g_pytest.txt
echo "--- ept4"; timeout 300 python tools/ckks_ops_bench.py 2>&1 | grep "|"
echo "--- no ept4"; PHA_LIB_OVERRIDE=$R/phantom-fhe_amd/phantom_fhe_amd/libphantom_amd_noept4.so timeout 300 python tools/ckks_ops_bench.py 2>&1 | grep "|"
echo "--- ept4 N=2^15"; PHA_OPS_LOGN=15 timeout 300 python tools/ckks_ops_bench.py 2>&1 | grep "|"
echo "--- no ept4 N=2^15"; PHA_OPS_LOGN=15 PHA_LIB_OVERRIDE=$R/phantom-fhe_amd/phantom_fhe_amd/libphantom_amd_noept4.so timeout 300 python tools/ckks_ops_bench.py 2>&1 | grep "|"
TAG=r03g_ks bash tools/ks_trace.sh | grep ntt_pass
