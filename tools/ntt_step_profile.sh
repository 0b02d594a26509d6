#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
trace() {  # name
  rm -rf /tmp/prof_$1
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_$1 -o trace -- python $R/tools/ntt_step_only.py > /tmp/prof_$1.log 2>&1
  python - <<PY
import sys
sys.path.insert(0, "$R/tools")
import summarize_prof as S
S.by_grid("/tmp/prof_$1", "$OUT/${TAG:-r04j}_bygrid_$1.csv")
print("== $1"); print("".join(l for l in open("$OUT/${TAG:-r04j}_bygrid_$1.csv") if "ntt_pass" in l))
PY
}
trace product
PHA_LIB_OVERRIDE=$R/phantom-fhe_amd/phantom_fhe_amd/libphantom_amd_nocomp1.so trace nocomp1
run() { timeout 200 rocprofv3 --pmc $2 --output-format csv -d /tmp/pmc_$1 -o pmc -- python $R/tools/ntt_step_only.py > /tmp/pmc_$1.log 2>&1 || echo "pass $1 failed or timed out"; }
run a "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU"
run b "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
run c "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum"
python - <<PY
import sys
sys.path.insert(0, "$R/tools")
import summarize_prof as S
for n in "abc":
    S.pmc("/tmp/pmc_" + n, "$OUT/${TAG:-r04j}_pmc_" + n + ".csv")
PY
grep -h "ntt_pass" $OUT/${TAG:-r04j}_pmc_*.csv | sed 's/void ntt_pass_kernel//; s/(NttKArgs)//' | cut -d, -f1,7,8,9- | sort
