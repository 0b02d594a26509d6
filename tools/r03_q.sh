#!/bin/bash
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_rns.py tests/test_gpu_fuzz.py tests/test_gpu_pyphantom.py -q -m gpu -x > $O/r03q_pytest.txt 2>&1
tail -4 $O/r03q_pytest.txt
timeout 200 python tools/time_bfv_mul.py 2>&1 | grep BFV | tee $O/r03q_bfvmul.txt
