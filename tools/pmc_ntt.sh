#!/bin/bash
# NTT stall analysis (GPU box): PMC passes over tools/ntt_only.py, summarised per kernel and grid.  Every pass is
# bounded by `timeout` (a counter set that the profiler cannot schedule must not eat the GPU budget).
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() { timeout 100 rocprofv3 --pmc $2 --output-format csv -d /tmp/pmc_$1 -o pmc -- python $R/tools/ntt_only.py > /tmp/pmc_$1.log 2>&1 || echo "pass $1 failed or timed out"; }
run a "SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY"
run c "TA_BUSY TA_ADDR_STALLED_BY_TC_CYCLES TA_DATA_STALLED_BY_TC_CYCLES TA_ADDR_STALLED_BY_TD_CYCLES TA_TOTAL_WAVEFRONTS GRBM_GUI_ACTIVE"
python - <<PY
import sys
sys.path.insert(0, "$R/tools")
import summarize_prof as S
for n in "ac":
    S.pmc("/tmp/pmc_" + n, "$OUT/${TAG:-r01}_ntt_pmc_" + n + ".csv")
PY
grep -h "ntt_pass" $OUT/${TAG:-r01}_ntt_pmc_*.csv | sed 's/void ntt_pass_kernel//; s/(NttKArgs)//' | cut -d, -f1,7,8,9- | sort
