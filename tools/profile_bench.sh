#!/bin/bash
# Runs on the GPU box: bench line + rocprofv3 kernel trace + PMC passes (separate runs, as the
# MI355X guide prescribes).  Outputs under gpurun_out/.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 python $R/bench.py > $OUT/bench.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_trace -o trace -- python $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline > $OUT/prof_trace.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d $OUT/prof_pmc_sq -o pmc -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline > $OUT/prof_pmc_sq.log 2>&1
timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/prof_pmc_lds -o pmc -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline > $OUT/prof_pmc_lds.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/prof_pmc_fetch -o pmc -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline > $OUT/prof_pmc_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/prof_pmc_write -o pmc -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline > $OUT/prof_pmc_write.log 2>&1
python $R/tools/summarize_prof.py $OUT ${TAG:-r02}
find $OUT -name "*.csv" | head -30; rm -rf $OUT/prof_trace $OUT/prof_pmc_sq $OUT/prof_pmc_lds $OUT/prof_pmc_fetch $OUT/prof_pmc_write
ls -la $OUT; du -sh $OUT
