#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests -x -q -m gpu > $OUT/r03c_pytest.txt 2>&1
tail -8 $OUT/r03c_pytest.txt
timeout 900 python bench.py > $OUT/r03c_bench.json 2> $OUT/r03c_bench.err
tail -c 1500 $OUT/r03c_bench.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r03c_bench.json"))
print("NTT/s", d["value"], "ms/step", d["ms_per_step"], "frac", d["roofline"]["frac"], "copy", d["roofline"]["calibrated_copy_GBps"], d["roofline"]["torch_copy_GBps"])
h=d["hommul_relin_rescale"]; print("hommul", h["ms_per_op"], h["gpu_ms_per_op"]["mean_ms"], "batched", h["batched"]["ms_per_op"])
print("c4", d["keyswitch_c4"]["value"], d["keyswitch_c4"]["checksum"])
print("c5", d["matvec_c5"])
PY
