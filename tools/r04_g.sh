#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
T0=$(date +%s)
timeout 1500 python -m pytest tests -x -q -m gpu > $OUT/r04g_pytest.txt 2>&1
grep -E "passed|failed|error" $OUT/r04g_pytest.txt | tail -3
echo "pytest seconds: $(( $(date +%s) - T0 ))"
python tools/exp_variants.py product 2>&1 | tee $OUT/r04g_product.txt
python bench.py --steps 20 --warmup 5 --no-c5 > $OUT/r04g_bench.json 2> $OUT/r04g_bench.err; tail -c 300 $OUT/r04g_bench.err
python - <<PY
import json
d=json.load(open("gpurun_out/r04g_bench.json"))
r=d["roofline"]
print("NTT/s", d["value"], "ms/step", d["ms_per_step"], "frac", r["frac"], "own copy", r["calibrated_copy_GBps"], "torch copy", r["torch_copy_GBps"])
print("single", d["single_polynomial"]["mall_resident"]["mean_ms"], d["single_polynomial"]["hbm_resident"]["mean_ms"])
h=d["hommul_relin_rescale"]; print("hommul", h["ms_per_op"], h["gpu_ms_per_op"]["mean_ms"], "3-launcher", h["three_launcher_sequence_gpu_ms_per_op"]["mean_ms"], "batched", h["batched"]["ms_per_op"])
print("c4", d["keyswitch_c4"]["value"])
PY
