#!/bin/bash
# MFMA modular GEMM: parity + timing
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_rns.py -q -m gpu -k gemm -x > $O/r03n_pytest.txt 2>&1
tail -15 $O/r03n_pytest.txt
timeout 120 python tools/time_gemm.py > $O/r03n_gemm.txt 2>&1
cat $O/r03n_gemm.txt
