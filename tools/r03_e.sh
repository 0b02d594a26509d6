#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests/test_gpu_workloads.py -x -q -m gpu -k "bsgs" > $OUT/r03e_pytest.txt 2>&1
tail -4 $OUT/r03e_pytest.txt
timeout 300 python tools/time_bsgs.py 16x8 32x4 64x2 2>&1 | grep steps
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_trace -o trace -- python $R/tools/time_bsgs.py 32x4 > $OUT/prof_trace.log 2>&1
python $R/tools/summarize_prof.py $OUT r03e > /dev/null 2>&1
rm -rf $OUT/prof_trace
grep -v "at::\|rocclr" $OUT/r03e_kernel_by_grid.csv | cut -c1-150
