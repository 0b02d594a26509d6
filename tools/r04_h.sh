#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
L=$R/phantom-fhe_amd/phantom_fhe_amd
timeout 900 python -m pytest tests/test_gpu_ntt_variants.py tests/test_gpu_reference_checks.py tests/test_gpu_bench.py -x -q -m gpu 2>&1 | tail -5
echo "== small launches 2^16: product"; python tools/time_small_ntt.py 2>&1 | grep -v amdgpu
echo "== small launches 2^16: noept4"; PHA_LIB_OVERRIDE=$L/libphantom_amd_noept4.so python tools/time_small_ntt.py 2>&1 | grep -v amdgpu
echo "== small launches 2^15: product"; PHA_OPS_LOGN=15 python tools/time_small_ntt.py 2>&1 | grep -v amdgpu
echo "== small launches 2^15: split15"; PHA_OPS_LOGN=15 PHA_LIB_OVERRIDE=$L/libphantom_amd_split15.so python tools/time_small_ntt.py 2>&1 | grep -v amdgpu
echo "== c4 product"; python tools/time_c4.py 2>&1 | grep -v amdgpu | head -3
echo "== c4 split15"; PHA_LIB_OVERRIDE=$L/libphantom_amd_split15.so python tools/time_c4.py 2>&1 | grep -v amdgpu | head -3
echo "== parity of split15 at C4 (test_gpu_rns keyswitch stages + workloads config4)"
PHA_LIB_OVERRIDE=$L/libphantom_amd_split15.so timeout 600 python -m pytest tests/test_gpu_workloads.py -x -q -m gpu -k "config4" 2>&1 | tail -2
