#!/bin/bash
# r06: SQ counters of the batched modular GEMM (30 x 256^3, 50-bit moduli), one build per pass (PHA_LIB_OVERRIDE), own PMC runs
# (rocprofv3 --pmc with --kernel-trace only).  Prints per kernel: calls, mean duration, and the counters per dispatch.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
L=$R/phantom-fhe_amd/phantom_fhe_amd
cd /tmp && export TMPDIR=/tmp
SETS=("SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU"
      "SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU"
      "GRBM_GUI_ACTIVE SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA")
for name in ${VARIANTS:-g1pass product}; do
  if [ $name = product ]; then unset PHA_LIB_OVERRIDE; else export PHA_LIB_OVERRIDE=$L/libphantom_amd_$name.so; fi
  i=0
  for set in "${SETS[@]}"; do
    timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/gp_${name}_$i -o pmc -- python $R/tools/time_gemm.py > $OUT/gp_${name}_$i.log 2>&1
    i=$((i+1))
  done
done
unset PHA_LIB_OVERRIDE
python - <<'PY'
import csv, glob, os, collections
out = os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out"
for name in os.environ.get("VARIANTS", "g1pass product").split():
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.Counter()
    dur = collections.defaultdict(list)
    for i in range(3):
        for f in glob.glob(f"{out}/gp_{name}_{i}/**/*counter_collection*.csv", recursive=True):
            seen = set()
            for r in csv.DictReader(open(f)):
                k = r["Kernel_Name"].split("(")[0][:70]
                if "gemm" not in k:
                    continue
                agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
                key = (k, r["Dispatch_Id"])
                if i == 0 and key not in seen:
                    seen.add(key)
                    cnt[k] += 1
        for f in glob.glob(f"{out}/gp_{name}_{i}/**/*kernel_trace*.csv", recursive=True):
            if i:
                continue
            for r in csv.DictReader(open(f)):
                k = r["Kernel_Name"].split("(")[0][:70]
                if "gemm" in k:
                    dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for k in agg:
        n = max(cnt[k], 1)
        d = dur.get(k, [0])
        print(f"== {name}: {k}  dispatches {n}  mean {sum(d)/max(len(d),1):.1f} us (under the profiler)")
        for c, v in sorted(agg[k].items()):
            print(f"   {c:28s} {v / n:16.1f} per dispatch")
PY
rm -rf $OUT/gp_*/
