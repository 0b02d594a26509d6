#!/bin/bash
# r06: A/B of the GEMM builds in one session: parity tests under each variant library, then timing (tools/time_gemm.py, 50-bit row)
cd ${GRAFT_REPO_ROOT:-/root/repo}
L=$PWD/phantom-fhe_amd/phantom_fhe_amd
for name in ${VARIANTS:-product g3}; do
  if [ $name = product ]; then unset PHA_LIB_OVERRIDE; else export PHA_LIB_OVERRIDE=$L/libphantom_amd_$name.so; fi
  printf "%-8s tests: " $name; python -m pytest tests/test_gpu_rns.py tests/test_gpu_reference_checks.py -q -k "gemm or matmul or reference" 2>&1 | tail -1
done
for rep in 1 2 3; do
for name in ${VARIANTS:-product g3}; do
  if [ $name = product ]; then unset PHA_LIB_OVERRIDE; else export PHA_LIB_OVERRIDE=$L/libphantom_amd_$name.so; fi
  printf "%-8s " $name; python tools/time_gemm.py 2>/dev/null | head -1
done
done
unset PHA_LIB_OVERRIDE
