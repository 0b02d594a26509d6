#!/bin/bash
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
for y in 1 2 3 4; do
  L=phantom-fhe_amd/phantom_fhe_amd/libphantom_amd_gy$y.so
  [ -f $L ] && { echo "Y=$y"; PHA_LIB_OVERRIDE=$PWD/$L GEMM_BATCH=30 timeout 120 python tools/gemm_stamps.py 2>&1 | grep -A2 "^tiles"; }
done | tee $O/r03o_stamps.txt
timeout 120 python tools/time_gemm.py 2>&1 | grep moduli | tee $O/r03o_gemm.txt
