#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
cd $R
L=$R/phantom-fhe_amd/phantom_fhe_amd
echo "== c4 product"; python tools/time_c4.py 2>&1 | grep -v amdgpu | head -4
echo "== c4 nozloop"; PHA_LIB_OVERRIDE=$L/libphantom_amd_nozloop.so python tools/time_c4.py 2>&1 | grep -v amdgpu | head -4
echo "== sweep 2^14..2^17 product"; SWEEP_LOGNS=14,15,17 python tools/ntt_sweep.py 2>/dev/null | grep "| 60 \|60 |" | tail -12
echo "== sweep nozloop"; PHA_LIB_OVERRIDE=$L/libphantom_amd_nozloop.so SWEEP_LOGNS=14,15,17 python tools/ntt_sweep.py 2>/dev/null | tail -12
timeout 1500 python -m pytest tests -x -q -m gpu > $OUT/r04l_pytest.txt 2>&1
grep -E "passed|failed|error" $OUT/r04l_pytest.txt | tail -3
