#!/bin/bash
# r05: what a 5-second utilisation sampler sees while the driver's own bench command runs (rocm-smi --showuse every 5 s in the
# background, the bench in the foreground); the sampler is stopped by its PID
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
cd $R
( while true; do echo "t=$(date +%s.%N | cut -c1-14) $(rocm-smi --showuse 2>/dev/null | grep -i 'GPU use' | head -1)"; sleep 5; done ) > $OUT/r06_gpu_busy_samples.txt &
SAMPLER=$!
T0=$(date +%s.%N | cut -c1-14)
python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/r06_gpu_busy_bench.json 2>/dev/null
T1=$(date +%s.%N | cut -c1-14)
kill $SAMPLER
echo "bench ran from t=$T0 to t=$T1"
cat $OUT/r06_gpu_busy_samples.txt
