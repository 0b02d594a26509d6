#!/bin/bash
# MFMA modular GEMM: parity + timing (+ timing-only variants)
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_rns.py -q -m gpu -k gemm -x > $O/r03n_pytest.txt 2>&1
tail -3 $O/r03n_pytest.txt
timeout 120 python tools/time_gemm.py > $O/r03n_gemm.txt 2>&1
cat $O/r03n_gemm.txt
for nb in $GEMM_BATCHES; do GEMM_BATCH=$nb timeout 120 python tools/time_gemm.py 2>&1 | grep "50-bit" | tee -a $O/r03n_gemm.txt; done
for x in 1 2 3 4 6 7 8; do
  L=phantom-fhe_amd/phantom_fhe_amd/libphantom_amd_gx$x.so
  [ -f $L ] && { echo "variant $x"; PHA_LIB_OVERRIDE=$PWD/$L timeout 120 python tools/time_gemm.py 2>&1 | grep moduli; } | tee -a $O/r03n_gemm.txt
done
L=phantom-fhe_amd/phantom_fhe_amd/libphantom_amd_gx5.so
[ -f $L ] && { PHA_LIB_OVERRIDE=$PWD/$L timeout 120 python tools/gemm_stamps.py 2>&1; PHA_LIB_OVERRIDE=$PWD/$L GEMM_BATCH=4 timeout 120 python tools/gemm_stamps.py 2>&1; } | tee $O/r03n_stamps.txt
