#!/bin/bash
# first GPU pass of round 3: parity of the new fused entry + NTT files, ops timing, key-switch kernel trace, bench line
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_rns.py tests/test_gpu_ntt.py tests/test_gpu_ntt_variants.py -x -q -m gpu > $OUT/r03a_pytest.txt 2>&1
tail -5 $OUT/r03a_pytest.txt
timeout 300 python tools/ckks_ops_bench.py > $OUT/r03a_ckks_ops.md 2>&1
cat $OUT/r03a_ckks_ops.md
TAG=r03a_ks bash tools/ks_trace.sh
timeout 600 python bench.py > $OUT/r03a_bench.json 2> $OUT/r03a_bench.err
tail -c 600 $OUT/r03a_bench.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r03a_bench.json"))
print("NTT/s", d["value"], "ms/step", d["ms_per_step"], "frac", d["roofline"]["frac"], "copy", d["roofline"]["calibrated_copy_GBps"], d["roofline"]["torch_copy_GBps"])
h=d["hommul_relin_rescale"]; print("hommul", h["ms_per_op"], h["gpu_ms_per_op"], "3-launcher", h["three_launcher_sequence_gpu_ms_per_op"], "batched", h["batched"]["ms_per_op"])
print("c4", d["keyswitch_c4"]["value"])
PY
