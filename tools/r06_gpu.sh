#!/bin/bash
# The GPU-box half of tools/r06_final.sh (run it through that script: it builds first and records what it built).
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
TAG=${TAG:-r06z}
mkdir -p $OUT
cd $R
# the libraries this box loads are the ones the build container recorded
if [ -f profiles/${TAG}_build.txt ]; then
  grep -E "\.so$" profiles/${TAG}_build.txt | sha256sum -c --quiet - && echo "build check: the .so files match profiles/${TAG}_build.txt" || { echo "build check FAILED"; exit 1; }
fi
T0=$(date +%s)
timeout 1500 python -m pytest tests -x -q -m gpu > $OUT/${TAG}_pytest.txt 2>&1
grep -E "passed|failed" $OUT/${TAG}_pytest.txt | tail -2
echo "pytest seconds: $(( $(date +%s) - T0 ))"
T0=$(date +%s)
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py --preflight > $OUT/${TAG}_preflight.json 2>/dev/null
PHA_BENCH_FORCE_DIST=1 python bench.py --preflight > $OUT/${TAG}_preflight_rccl1.json 2>/dev/null
bash tools/profile_r06.sh
echo "profile seconds: $(( $(date +%s) - T0 ))"
# the reference's benchmark shapes on the same tree (benchmark/ntt_bench.cu:104-117, keyswitch_bench.cu:16-34, ckks_bench.cu:168-205)
T0=$(date +%s)
timeout 600 python tools/ntt_sweep.py > $OUT/${TAG}_ntt_sweep.md 2>/dev/null
timeout 600 python tools/keyswitch_sweep.py > $OUT/${TAG}_keyswitch_sweep.md 2>/dev/null
timeout 600 python tools/ckks_ops_bench.py > $OUT/${TAG}_ckks_ops.md 2>/dev/null
echo "sweep seconds: $(( $(date +%s) - T0 ))"
cat $OUT/${TAG}_stages.txt
cat $OUT/${TAG}_stages_batched.txt
tail -c 400 $OUT/${TAG}_bench.err
python - <<PY
import json
d=json.load(open("gpurun_out/${TAG}_bench_full.json"))
r=d["roofline"]
print("NTT/s", d["value"], "ms/step", d["ms_per_step"], "frac", r["frac"], "sustained", r["sustained"]["median_ms_per_step"], "own copy", r["calibrated_copy_GBps"], "ceil", r["ceiling_two_pass"], r["frac_of_ceiling"])
print("traffic", r["traffic"], "single", d["single_polynomial"]["mall_resident"]["mean_ms"], d["single_polynomial"]["hbm_resident"]["mean_ms"])
h=d["hommul_relin_rescale"]; b=h["batched"]
print("hommul", h["ms_per_op"], h["gpu_ms_per_op"]["mean_ms"], "3-launcher", h["three_launcher_sequence_gpu_ms_per_op"]["mean_ms"], "traffic ratio", h.get("traffic_ratio"))
print("batched", b["ms_per_op"], "B", b["batch"], "fixed 8", b["fixed_batch_8"], "sustained", b["sustained"], "traffic ratio", b.get("traffic_ratio"))
print("c4", d["keyswitch_c4"]["value"], "c5", d["matvec_c5"]["ms_per_block"], d["matvec_c5"]["value"])
print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["all_cores"]["value"], d["cpu_baseline"]["all_cores"]["cores"])
PY
