#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_rns.py -x -q -m gpu -k "keyswitch or rescale or hommul" 2>&1 | grep -E "passed|failed" | tail -1
for v in "" ip5; do
  echo "=== ${v:-w4 (product)}"
  if [ -n "$v" ]; then export PHA_LIB_OVERRIDE=$R/phantom-fhe_amd/phantom_fhe_amd/libphantom_amd_$v.so; else unset PHA_LIB_OVERRIDE; fi
  timeout 300 python tools/ckks_ops_bench.py 2>&1 | grep -E "relinearize|multiply \+"
  PHA_OPS_LOGN=15 timeout 300 python tools/ckks_ops_bench.py 2>&1 | grep -E "relinearize \(|multiply \+"
done
unset PHA_LIB_OVERRIDE
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_trace -o trace -- python $R/tools/ckks_ops_bench.py > $OUT/ks_trace.log 2>&1
python $R/tools/summarize_prof.py $OUT r03k > /dev/null 2>&1
rm -rf $OUT/prof_trace
grep -E "modup_ip" $OUT/r03k_kernel_by_grid.csv | cut -c1-150
