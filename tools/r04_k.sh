#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_stage
timeout 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_stage -o trace -- python $R/tools/traffic_probe.py hommul > /tmp/prof_stage.log 2>&1
python $R/tools/stage_table.py /tmp/prof_stage $OUT/r04k_stages.json 2>&1 | tee $OUT/r04k_stages.txt
