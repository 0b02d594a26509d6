#!/bin/bash
# r05 first GPU session: the new tests, the small-launch one-launch experiment, the batched HomMul baseline records, one bench line
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
TAG=${TAG:-r05a}
mkdir -p $OUT
cd $R
T0=$(date +%s)
timeout 600 python -m pytest tests/test_gpu_abi_rows.py "tests/test_gpu_bench.py::test_bench_line_contract" "tests/test_gpu_bench.py::test_bench_preflight" -x -q -m gpu > $OUT/${TAG}_pytest.txt 2>&1
tail -5 $OUT/${TAG}_pytest.txt
echo "pytest seconds: $(( $(date +%s) - T0 ))"
T0=$(date +%s)
PHA_LIB_OVERRIDE=$R/phantom-fhe_amd/phantom_fhe_amd/libphantom_amd_exp.so timeout 300 python tools/exp_onelaunch_small.py > $OUT/${TAG}_onelaunch_small.txt 2>&1
cat $OUT/${TAG}_onelaunch_small.txt | tail -40
echo "onelaunch seconds: $(( $(date +%s) - T0 ))"
T0=$(date +%s)
cd /tmp && export TMPDIR=/tmp
for B in 8 32; do
  rm -rf /tmp/prof_hb$B
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_hb$B -o trace -- python $R/tools/traffic_probe.py hommul_batched:$B > $OUT/${TAG}_hb$B.log 2>&1
done
python $R/tools/stage_table_batched.py $OUT/stages_batched.json 8=/tmp/prof_hb8 32=/tmp/prof_hb32 > $OUT/${TAG}_stages_batched.txt 2>&1
cat $OUT/${TAG}_stages_batched.txt
echo "stage seconds: $(( $(date +%s) - T0 ))"
T0=$(date +%s)
cd $R
timeout 600 python bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
tail -c 300 $OUT/${TAG}_bench.err
python - <<PY
import json
d=json.load(open("$OUT/${TAG}_bench.json"))
r=d["roofline"]; print("NTT", d["value"], d["ms_per_step"], r["frac"], "sustained", r["sustained"])
h=d["hommul_relin_rescale"]; print("hommul", h["ms_per_op"], h["gpu_ms_per_op"]["mean_ms"], "batched", h["batched"]["ms_per_op"], h["batched"]["batch"], [ (e["batch"], round(e["ms_per_op"],4)) for e in h["batched"]["sweep"]], h["batched"]["sustained"])
print("cpu", d["cpu_baseline"]["value"])
PY
echo "bench seconds: $(( $(date +%s) - T0 ))"
