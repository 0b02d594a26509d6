#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
python tools/exp_variants.py product ept4 hoistC1 hoistC2 hoistS1 s64a s64ah2 s64aS6 occS6 product 2>&1 | tee $OUT/r04e_variants.txt
echo "== chunks with s64a"
PHA_LIB_OVERRIDE=$R/phantom-fhe_amd/phantom_fhe_amd/libphantom_amd_s64a.so python tools/exp_mall_chunks.py 2>&1 | grep -v amdgpu.ids | head -10 | tee $OUT/r04e_chunks_s64a.txt
