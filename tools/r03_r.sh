#!/bin/bash
# MFMA base conversion: parity of everything that converts, then the HomMul / config-4 numbers
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_rns.py tests/test_gpu_fuzz.py tests/test_gpu_workloads.py -q -m gpu -x > $O/r03r_pytest.txt 2>&1
tail -4 $O/r03r_pytest.txt
timeout 600 python bench.py --steps 30 --warmup 5 --no-c5 --no-cpu-baseline > $O/r03r_bench.json 2> $O/r03r_bench.err
python - <<PY
import json
d=json.load(open("gpurun_out/r03r_bench.json"))
h=d["hommul_relin_rescale"]; print("hommul wall", h["ms_per_op"], "gpu", h["gpu_ms_per_op"]["mean_ms"], "batched", h["batched"]["ms_per_op"], "c4", d["keyswitch_c4"]["value"])
PY
timeout 300 python tools/time_bfv_mul.py 2>&1 | grep BFV
