#!/bin/bash
# r05: PMC traffic of the batched op (5 op sets of 8) for the product and for an experiment build (default: mcsoff = the separate
# conversion + strided pass), FETCH_SIZE and WRITE_SIZE in separate runs; prints bytes per op for each
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for name in product ${VARIANTS:-mcsoff}; do
  if [ $name = product ]; then unset PHA_LIB_OVERRIDE; else export PHA_LIB_OVERRIDE=$R/phantom-fhe_amd/phantom_fhe_amd/libphantom_amd_$name.so; fi
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/tr_$c
    timeout 300 rocprofv3 --pmc $c --output-format csv -d /tmp/tr_$c -o pmc -- python $R/tools/traffic_probe.py hommul_batched:8 > /tmp/tr.log 2>&1
  done
  python - <<PY
import csv, glob
tot={}
for c in ("FETCH_SIZE","WRITE_SIZE"):
    s=0.0
    for f in glob.glob(f"/tmp/tr_{c}/**/*counter_collection*.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "pha::" in r["Kernel_Name"]: s+=float(r["Counter_Value"])
    tot[c]=s
b=(2.0*tot["FETCH_SIZE"]+tot["WRITE_SIZE"])*1024.0/40
print("$name: batched HomMul + relinearize + rescale, B = 8: %.1f MB per op = %.3f x 974.1 MB (fetch %.1f MB, write %.1f MB)" % (b/1e6, b/974127104.0, 2*tot["FETCH_SIZE"]*1024/40/1e6, tot["WRITE_SIZE"]*1024/40/1e6))
PY
done | tee $OUT/r05_traffic_ab.txt
