#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
T0=$(date +%s)
timeout 1500 python -m pytest tests -x -q -m gpu > $OUT/${TAG:-r04z}_pytest.txt 2>&1
grep -E "passed|failed" $OUT/${TAG:-r04z}_pytest.txt | tail -2
echo "pytest seconds: $(( $(date +%s) - T0 ))"
T0=$(date +%s)
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
bash tools/profile_r04.sh
echo "profile seconds: $(( $(date +%s) - T0 ))"
# the reference's benchmark shapes on the same tree (benchmark/ntt_bench.cu:104-117, keyswitch_bench.cu:16-34, ckks_bench.cu:168-205)
T0=$(date +%s)
timeout 600 python tools/ntt_sweep.py > $OUT/${TAG:-r04z}_ntt_sweep.md 2>/dev/null
timeout 600 python tools/keyswitch_sweep.py > $OUT/${TAG:-r04z}_keyswitch_sweep.md 2>/dev/null
timeout 600 python tools/ckks_ops_bench.py > $OUT/${TAG:-r04z}_ckks_ops.md 2>/dev/null
echo "sweep seconds: $(( $(date +%s) - T0 ))"
cat $OUT/${TAG:-r04z}_stages.txt
tail -c 400 $OUT/${TAG:-r04z}_bench.err
python - <<PY
import json
d=json.load(open("gpurun_out/${TAG:-r04z}_bench.json"))
r=d["roofline"]
print("NTT/s", d["value"], "ms/step", d["ms_per_step"], "frac", r["frac"], "own copy", r["calibrated_copy_GBps"], "torch copy", r["torch_copy_GBps"], "ceil", r["ceiling_two_pass"], r["frac_of_ceiling"])
print("single", d["single_polynomial"]["mall_resident"]["mean_ms"], d["single_polynomial"]["hbm_resident"]["mean_ms"])
h=d["hommul_relin_rescale"]; print("hommul", h["ms_per_op"], h["gpu_ms_per_op"]["mean_ms"], "3-launcher", h["three_launcher_sequence_gpu_ms_per_op"]["mean_ms"], "batched", h["batched"]["ms_per_op"])
print("c4", d["keyswitch_c4"]["value"], "c5", d["matvec_c5"]["ms_per_block"], d["matvec_c5"]["value"])
print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["all_cores"]["value"], d["cpu_baseline"]["all_cores"]["cores"])
PY
