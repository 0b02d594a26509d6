#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
python tools/exp_variants.py product knobs occS6 occS8 occC7 occC8 s128 s64a s64b knobs:PHA_X_ZRUN=16 knobs:PHA_X_ZRUN=32 knobs:PHA_X_ZRUN=64 knobs:PHA_X_ZRUN=128 s64a:PHA_X_ZRUN=32 s64b:PHA_X_ZRUN=32 product 2>&1 | tee $OUT/r04d_variants.txt
for v in product s128 s64a s64b occS6; do
  if [ $v = product ]; then unset PHA_LIB_OVERRIDE; else export PHA_LIB_OVERRIDE=$R/phantom-fhe_amd/phantom_fhe_amd/libphantom_amd_$v.so; fi
  rm -rf /tmp/prof_$v
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_$v -o trace -- python $R/tools/ntt_step_only.py > /tmp/prof_$v.log 2>&1)
  python - <<PY
import sys
sys.path.insert(0, "$R/tools")
import summarize_prof as S
S.by_grid("/tmp/prof_$v", "$OUT/r04d_bygrid_$v.csv")
print("== $v"); print(open("$OUT/r04d_bygrid_$v.csv").read())
PY
done
