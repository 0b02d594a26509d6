#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
python tools/exp_variants.py product nt15 nt5 nt10 split splitnt nozfast knobs knobs:PHA_X_LDS_C=8192 knobs:PHA_X_LDS_C=16384 knobs:PHA_X_LDS_C=28000 knobs:PHA_X_LDS_S=21000 knobs:PHA_X_LDS_S=48000 product 2>&1 | tee $OUT/r04b_variants.txt
