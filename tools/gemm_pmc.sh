#!/bin/bash
# MFMA modular GEMM under the profiler: kernel trace + two counter passes (matrix-pipe busy, LDS) of tools/time_gemm.py
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; TAG=${TAG:-r03w}; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/prof_trace $OUT/prof_pmc_sq $OUT/prof_pmc_lds
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_trace -o trace -- python $R/tools/time_gemm.py > $OUT/${TAG}_gemm.log 2>&1
timeout 200 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU --output-format csv -d $OUT/prof_pmc_sq/gemm -o pmc -- python $R/tools/time_gemm.py >> $OUT/${TAG}_gemm.log 2>&1
timeout 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $OUT/prof_pmc_lds/gemm -o pmc -- python $R/tools/time_gemm.py >> $OUT/${TAG}_gemm.log 2>&1
cd $R && python tools/summarize_prof.py $OUT ${TAG}_gemm
rm -rf $OUT/prof_trace $OUT/prof_pmc_sq $OUT/prof_pmc_lds
grep -h "gemm_mfma" $OUT/${TAG}_gemm_kernel_stats.csv $OUT/${TAG}_gemm_pmc_sq.csv $OUT/${TAG}_gemm_pmc_lds.csv | cut -c1-200
tail -3 $OUT/${TAG}_gemm.log
