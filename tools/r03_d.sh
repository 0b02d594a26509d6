#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests/test_gpu_workloads.py -x -q -m gpu > $OUT/r03d_pytest.txt 2>&1
tail -8 $OUT/r03d_pytest.txt
timeout 900 python bench.py --steps 20 --no-cpu-baseline > $OUT/r03d_bench.json 2> $OUT/r03d_bench.err
tail -c 800 $OUT/r03d_bench.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r03d_bench.json"))
print("c4", d["keyswitch_c4"]["value"], d["keyswitch_c4"]["checksum"])
print("c5", d["matvec_c5"])
PY
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_trace -o trace -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/prof_trace.log 2>&1
python $R/tools/summarize_prof.py $OUT r03d > /dev/null 2>&1
rm -rf $OUT/prof_trace
grep -E "bsgs|multi|hoist|galois" $OUT/r03d_kernel_by_grid.csv | cut -c1-160
