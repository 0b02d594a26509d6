#!/bin/bash
# per-kernel GPU times of the key switch / rescale at C3 (rocprofv3 kernel trace of tools/ckks_ops_bench.py), by (kernel, grid)
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_trace -o trace -- python $R/tools/ckks_ops_bench.py > $OUT/ks_trace.log 2>&1
python $R/tools/summarize_prof.py $OUT ${TAG:-ks} > /dev/null 2>&1
grep -E "bconv|inner_prod|ntt_pass|ew_kernel|modup_ip" $OUT/${TAG:-ks}_kernel_by_grid.csv | cut -c1-150
rm -rf $OUT/prof_trace
