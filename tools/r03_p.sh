#!/bin/bash
# f3 row: kernel trace of the BFV multiply variants (BEHZ / HPS) + relinearize at the C4 parameter set
R=/root/repo; OUT=$R/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/prof_trace
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_trace -o trace -- python $R/tools/time_bfv_mul.py > $OUT/r03p_bfvmul.log 2>&1
cd $R && python tools/summarize_prof.py $OUT r03p
tail -3 $OUT/r03p_bfvmul.log
rm -rf $OUT/prof_trace
