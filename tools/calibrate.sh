#!/bin/bash
# r04 calibration session: streaming ceiling (tools/stream_calib.hip), cross-lane exchange (tools/xlane_bench.hip), MALL-chunk experiment
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
timeout 300 tools/stream_calib > $OUT/r04_stream_calib.txt 2>&1; echo "calib rc $?"
timeout 120 tools/xlane_bench > $OUT/r04_xlane.txt 2>&1; echo "xlane rc $?"
cat $OUT/r04_xlane.txt
timeout 600 python tools/exp_mall_chunks.py > $OUT/r04_mall_chunks.txt 2>&1; echo "mall rc $?"
cat $OUT/r04_mall_chunks.txt
cat $OUT/r04_stream_calib.txt
