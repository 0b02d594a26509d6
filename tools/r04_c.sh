#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
timeout 300 tools/pattern_bench 2>&1 | tee $OUT/r04c_pattern_bench.txt
python tools/exp_variants.py product nocomp1 nocomp2 nocomp2nt 2>&1 | tee $OUT/r04c_nocompute.txt
