#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
for v in p6 p7; do
  echo "=== $v"
  PHA_LIB_OVERRIDE=$R/phantom-fhe_amd/phantom_fhe_amd/libphantom_amd_$v.so timeout 600 python -m pytest tests/test_gpu_ntt.py -x -q -m gpu 2>&1 | grep -E "passed|failed" | tail -1
  PHA_LIB_OVERRIDE=$R/phantom-fhe_amd/phantom_fhe_amd/libphantom_amd_$v.so timeout 300 python tools/ckks_ops_bench.py 2>&1 | grep -E "relinearize|multiply \+"
  PHA_OPS_LOGN=15 PHA_LIB_OVERRIDE=$R/phantom-fhe_amd/phantom_fhe_amd/libphantom_amd_$v.so timeout 300 python tools/ckks_ops_bench.py 2>&1 | grep -E "relinearize|multiply \+"
done
echo "=== p5 (product)"
timeout 300 python tools/ckks_ops_bench.py 2>&1 | grep -E "relinearize|multiply \+"
PHA_OPS_LOGN=15 timeout 300 python tools/ckks_ops_bench.py 2>&1 | grep -E "relinearize|multiply \+"
cd /tmp && export TMPDIR=/tmp
for v in p6 p7; do
PHA_LIB_OVERRIDE=$R/phantom-fhe_amd/phantom_fhe_amd/libphantom_amd_$v.so timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_trace -o trace -- python $R/tools/ckks_ops_bench.py > $OUT/ks_trace.log 2>&1
python $R/tools/summarize_prof.py $OUT r03h_$v > /dev/null 2>&1
rm -rf $OUT/prof_trace
echo "--- $v strided passes"; grep "true, [0-9], [0-9], [0-9], 4" $OUT/r03h_${v}_kernel_by_grid.csv | cut -c1-130
done
