#!/bin/bash
# Round-6 profile set (the r05 set + the driver's exact bench command with the line-size check), ONE command on the GPU box (TAG=r05x bash tools/profile_r06.sh): bench line, rocprofv3 kernel trace of the
# same command (by kernel and by grid), the per-stage table of one HomMul (profiles/stages.json) and, r05, of one op INSIDE A BATCH of 8
# and of 32 (profiles/stages_batched.json), PMC passes in their own runs (SQ / LDS counters over the two small workloads, FETCH_SIZE /
# WRITE_SIZE traffic of the step, the single op and the batched op -> profiles/traffic.json).  Everything lands in
# gpurun_out/ under the tag; copy the summaries into profiles/ (tools/collect_profiles.sh does that) and commit them WITH the
# tree they were measured on -- bench.py prints the sha of traffic.json / stages.json so a stale file shows.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
TAG=${TAG:-r06}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_trace -o trace -- python $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-c5 > $OUT/prof_trace.log 2>&1
timeout 200 rocprofv3 --kernel-trace --output-format csv -d $OUT/prof_stage -o trace -- python $R/tools/traffic_probe.py hommul > $OUT/prof_stage.log 2>&1
python $R/tools/stage_table.py $OUT/prof_stage $OUT/stages.json > $OUT/${TAG}_stages.txt 2>&1
for B in 8 32; do
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/prof_hb$B -o trace -- python $R/tools/traffic_probe.py hommul_batched:$B > $OUT/prof_hb$B.log 2>&1
done
python $R/tools/stage_table_batched.py $OUT/stages_batched.json 8=$OUT/prof_hb8 32=$OUT/prof_hb32 > $OUT/${TAG}_stages_batched.txt 2>&1
pmc() {   # name, counters...
  local name=$1; shift
  timeout 300 rocprofv3 --pmc "$@" --output-format csv -d $OUT/prof_pmc_$name/ntt -o pmc -- python $R/bench.py --only-ntt --steps 4 --warmup 1 --no-cpu-baseline --no-graph > $OUT/prof_pmc_$name.log 2>&1
  timeout 300 rocprofv3 --pmc "$@" --output-format csv -d $OUT/prof_pmc_$name/ops -o pmc -- python $R/tools/traffic_probe.py hommul >> $OUT/prof_pmc_$name.log 2>&1
  timeout 300 rocprofv3 --pmc "$@" --output-format csv -d $OUT/prof_pmc_$name/hb -o pmc -- python $R/tools/traffic_probe.py hommul_batched:8 >> $OUT/prof_pmc_$name.log 2>&1
}
if [ -z "$SKIP_PMC" ]; then
  pmc sq SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU
  pmc lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU GRBM_GUI_ACTIVE
fi
python $R/tools/summarize_prof.py $OUT $TAG
rm -rf $OUT/prof_trace $OUT/prof_stage $OUT/prof_hb8 $OUT/prof_hb32 $OUT/prof_pmc_sq $OUT/prof_pmc_lds
bash $R/tools/traffic.sh > $OUT/${TAG}_traffic.txt 2>&1
# the bench line LAST, with the records of THIS run in place: bench.py attaches profiles/{traffic,stages,stages_batched}.json with their sha, and
# tools/collect_profiles.sh copies the same three files from gpurun_out/ into profiles/ of the build container afterwards
cp $OUT/traffic.json $OUT/stages.json $OUT/stages_batched.json $R/profiles/ 2>/dev/null
# r06: the DRIVER'S OWN COMMAND; stdout must be ONE JSON line of at most 8000 bytes that json.loads; the full record
# (what r01-r05 printed) goes to the file named in the line
cd $R
timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 --full-out $OUT/${TAG}_bench_full.json > $OUT/${TAG}_bench_line.json 2> $OUT/${TAG}_bench.err
python3 - <<PY
import json
raw = open("$OUT/${TAG}_bench_line.json").read()
lines = [l for l in raw.splitlines() if l.strip()]
assert len(lines) == 1, f"{len(lines)} stdout lines"
d = json.loads(lines[-1])
assert len(lines[-1]) <= 8000, len(lines[-1])
for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "dtype", "config", "roofline", "cpu_baseline"):
    assert k in d, k
print("bench line:", len(lines[-1]), "bytes; value", d["value"], d["unit"], "ms_per_step", d["ms_per_step"], "frac", d["roofline"]["frac"])
PY
ls -la $OUT | grep -E "$TAG|traffic.json|stages.json|stages_batched.json"; du -sh $OUT
