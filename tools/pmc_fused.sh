#!/bin/bash
# HBM-side traffic and stall counters of the batched forward NTT: two launches (variant 1121) vs one launch (97)
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for v in 1121 97; do
  for c in FETCH_SIZE WRITE_SIZE "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
    tag=$(echo $c | cut -d' ' -f1)
    PHA_NTT_VARIANT=$v timeout 200 rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_${v}_$tag -o pmc -- python $R/bench.py --only-ntt --no-cpu-baseline --no-graph --steps 3 --warmup 1 > $OUT/pmc_${v}_$tag.log 2>&1
  done
done
python - <<'PY'
import csv, glob, os, collections
out = os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out"
for d in sorted(glob.glob(out + "/pmc_*/")):
    agg = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(d + "/**/*counter_collection*.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"]
            if "ntt_" not in name: continue
            grid = int(r["Grid_Size"]) // max(1, int(r["Workgroup_Size"]))
            if grid < 5000: continue
            kind = "fused" if "fused" in name else ("strided" if "Lb1E" in name.split("PassCfg")[1][:12] else "contig")
            k = (kind, grid, r["Counter_Name"], r.get("VGPR_Count"), r.get("Scratch_Size", ""))
            agg[k][0] += float(r["Counter_Value"]); agg[k][1] += 1
    for k, (s, n) in sorted(agg.items()):
        print(os.path.basename(d.rstrip("/")), k, "mean %.1f over %d" % (s / n, n))
PY
rm -rf $OUT/pmc_*/
