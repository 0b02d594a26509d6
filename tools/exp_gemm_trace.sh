#!/bin/bash
# r06: per-kernel durations of a GEMM build (VARIANT) under rocprofv3 --kernel-trace
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for name in ${VARIANTS:-g3}; do
  if [ $name = product ]; then unset PHA_LIB_OVERRIDE; else export PHA_LIB_OVERRIDE=$R/phantom-fhe_amd/phantom_fhe_amd/libphantom_amd_$name.so; fi
  rm -rf $R/gpurun_out/gt_$name
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/gt_$name -o t -- python $R/tools/time_gemm.py > /dev/null 2>&1
  echo "== $name"; f=$(find $R/gpurun_out/gt_$name -name "*kernel_stats.csv" | head -1); grep -i "gemm" $f | cut -c1-200
  rm -rf $R/gpurun_out/gt_$name
done
