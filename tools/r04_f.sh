#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
python tools/exp_variants.py product prioA prioB prioC prioD s64a s64aprioA s64aprioC product 2>&1 | tee $OUT/r04f_variants.txt
