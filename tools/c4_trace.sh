#!/bin/bash
# per-kernel GPU times of BASELINE config 4 (tools/time_c4.py: relinearize + rotate, N = 2^15, 30 + 15 limbs, batch 64)
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_trace -o trace -- python $R/tools/time_c4.py > $OUT/c4_trace.log 2>&1
python $R/tools/summarize_prof.py $OUT c4 > /dev/null 2>&1
sort -t, -k2 -n -r $OUT/c4_kernel_stats.csv | head -25 | cut -c1-170
rm -rf $OUT/prof_trace
