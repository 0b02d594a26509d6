#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
W=$R/phantom-fhe_amd/phantom_fhe_amd/libphantom_amd_wide.so
PHA_LIB_OVERRIDE=$W timeout 600 python -m pytest tests/test_gpu_ntt.py -x -q -m gpu 2>&1 | grep -E "passed|failed" | tail -1
for i in 1 2; do
python bench.py --only-ntt --no-cpu-baseline --steps 50 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('base', d['ms_per_step'], d['roofline']['frac'], d['single_polynomial']['hbm_resident']['mean_ms'])"
PHA_LIB_OVERRIDE=$W python bench.py --only-ntt --no-cpu-baseline --steps 50 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('wide', d['ms_per_step'], d['roofline']['frac'], d['single_polynomial']['hbm_resident']['mean_ms'])"
done
