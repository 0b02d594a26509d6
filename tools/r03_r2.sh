#!/bin/bash
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_rns.py -q -m gpu -x -k "keyswitch or modup or bconv" > $O/r03r_pytest.txt 2>&1
tail -2 $O/r03r_pytest.txt
timeout 600 python bench.py --steps 30 --warmup 5 --no-c5 --no-cpu-baseline > $O/r03r_bench.json 2> $O/r03r_bench.err
python - <<PY
import json
d=json.load(open("gpurun_out/r03r_bench.json"))
h=d["hommul_relin_rescale"]; print("hommul wall", h["ms_per_op"], "gpu", h["gpu_ms_per_op"]["mean_ms"], "batched", h["batched"]["ms_per_op"], "c4", d["keyswitch_c4"]["value"])
PY
cd /tmp; rm -rf /tmp/pst; timeout 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/pst -o trace -- python /root/repo/tools/traffic_probe.py hommul > /dev/null 2>&1
cd /root/repo; python tools/stage_table.py /tmp/pst 2>&1 | tail -12
