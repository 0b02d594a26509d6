#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
python tools/exp_variants.py product 2>&1 | tee $OUT/r04i_product.txt
T0=$(date +%s)
timeout 1500 python -m pytest tests -x -q -m gpu > $OUT/r04i_pytest.txt 2>&1
grep -E "passed|failed|error" $OUT/r04i_pytest.txt | tail -3
echo "pytest seconds: $(( $(date +%s) - T0 ))"
python tools/exp_variants.py product 2>&1 | tee -a $OUT/r04i_product.txt
