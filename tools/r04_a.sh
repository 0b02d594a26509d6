#!/bin/bash
# r04 session A: stream calibration, cross-lane exchange, MALL-chunk experiment, comm tests on the rebuilt library
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
timeout 300 tools/stream_calib > $OUT/r04a_stream_calib.txt 2>&1; echo "calib rc $?"
timeout 120 tools/xlane_bench > $OUT/r04a_xlane.txt 2>&1; echo "xlane rc $?"
cat $OUT/r04a_xlane.txt
timeout 600 python tools/exp_mall_chunks.py > $OUT/r04a_mall_chunks.txt 2>&1; echo "mall rc $?"
cat $OUT/r04a_mall_chunks.txt
timeout 600 python -m pytest tests/test_gpu_comm.py tests/test_gpu_ntt.py -x -q -m gpu 2>&1 | tail -3
cat $OUT/r04a_stream_calib.txt
