#!/bin/bash
# Runs on the GPU box: bench line + rocprofv3 kernel trace + PMC passes (separate runs, as the
# MI355X guide prescribes).  Outputs under gpurun_out/.
# The --pmc passes collect over two smaller workloads (bench.py --only-ntt = the headline kernels, tools/ckks_ops_bench.py =
# the key-switch / rescale kernels): counter collection over the whole bench (batched HomMul + config-4 legs) has
# crashed rocprofv3 on this image (SIGSEGV inside the tool), the kernel trace of the whole bench is fine.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 python $R/bench.py > $OUT/bench.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_trace -o trace -- python $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline > $OUT/prof_trace.log 2>&1
pmc() {   # name, counters...
  local name=$1; shift
  timeout 300 rocprofv3 --pmc "$@" --output-format csv -d $OUT/prof_pmc_$name/ntt -o pmc -- python $R/bench.py --only-ntt --steps 4 --warmup 1 --no-cpu-baseline --no-graph > $OUT/prof_pmc_$name.log 2>&1
  timeout 300 rocprofv3 --pmc "$@" --output-format csv -d $OUT/prof_pmc_$name/ops -o pmc -- python $R/tools/ckks_ops_bench.py >> $OUT/prof_pmc_$name.log 2>&1
}
pmc sq SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU
pmc lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU GRBM_GUI_ACTIVE
pmc fetch FETCH_SIZE
pmc write WRITE_SIZE
python $R/tools/summarize_prof.py $OUT ${TAG:-r02}
rm -rf $OUT/prof_trace $OUT/prof_pmc_sq $OUT/prof_pmc_lds $OUT/prof_pmc_fetch $OUT/prof_pmc_write
ls -la $OUT | grep ${TAG:-r02}; du -sh $OUT
