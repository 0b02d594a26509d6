#!/bin/bash
# Round-4 profile set (the r03 script under its new name), ONE command on the GPU box (TAG=r04x bash tools/profile_r04.sh): bench line, rocprofv3 kernel trace of the
# same command (by kernel and by grid), the per-stage table of one HomMul (profiles/stages.json), PMC passes in their own runs
# (SQ / LDS counters over the two small workloads, FETCH_SIZE / WRITE_SIZE traffic -> profiles/traffic.json).  Everything lands in
# gpurun_out/ under the tag; copy the summaries into profiles/ (tools/collect_profiles.sh does that) and commit them WITH the
# tree they were measured on -- bench.py prints the sha of traffic.json / stages.json so a stale file shows.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
TAG=${TAG:-r04}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 900 python $R/bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_trace -o trace -- python $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-c5 > $OUT/prof_trace.log 2>&1
timeout 200 rocprofv3 --kernel-trace --output-format csv -d $OUT/prof_stage -o trace -- python $R/tools/traffic_probe.py hommul > $OUT/prof_stage.log 2>&1
python $R/tools/stage_table.py $OUT/prof_stage $OUT/stages.json > $OUT/${TAG}_stages.txt 2>&1
pmc() {   # name, counters...
  local name=$1; shift
  timeout 300 rocprofv3 --pmc "$@" --output-format csv -d $OUT/prof_pmc_$name/ntt -o pmc -- python $R/bench.py --only-ntt --steps 4 --warmup 1 --no-cpu-baseline --no-graph > $OUT/prof_pmc_$name.log 2>&1
  timeout 300 rocprofv3 --pmc "$@" --output-format csv -d $OUT/prof_pmc_$name/ops -o pmc -- python $R/tools/traffic_probe.py hommul >> $OUT/prof_pmc_$name.log 2>&1
}
if [ -z "$SKIP_PMC" ]; then
  pmc sq SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU
  pmc lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU GRBM_GUI_ACTIVE
fi
python $R/tools/summarize_prof.py $OUT $TAG
rm -rf $OUT/prof_trace $OUT/prof_stage $OUT/prof_pmc_sq $OUT/prof_pmc_lds
bash $R/tools/traffic.sh > $OUT/${TAG}_traffic.txt 2>&1
ls -la $OUT | grep -E "$TAG|traffic.json|stages.json"; du -sh $OUT
