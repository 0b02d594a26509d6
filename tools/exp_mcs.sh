#!/bin/bash
# r05: A/B of the fused "base conversion + strided pass" mod-up (modup_conv_s1_kernel) against the separate kernels, in one session on
# one box: batched HomMul + relinearize + rescale us/op with an output checksum per build (tools/time_hommul_batched.py)
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
TAG=${TAG:-r05b}
mkdir -p $OUT
cd $R
T0=$(date +%s)
if [ -z "$SKIP_PYTEST" ]; then
timeout 900 python -m pytest tests/test_gpu_rns.py -x -q -m gpu > $OUT/${TAG}_pytest_rns.txt 2>&1
tail -4 $OUT/${TAG}_pytest_rns.txt
fi
echo "pytest seconds: $(( $(date +%s) - T0 ))"
T0=$(date +%s)
LIBDIR=$R/phantom-fhe_amd/phantom_fhe_amd
{
for rep in 1 2; do
  for name in product $VARIANTS; do
    if [ $name = product ]; then unset PHA_LIB_OVERRIDE; else export PHA_LIB_OVERRIDE=$LIBDIR/libphantom_amd_$name.so; fi
    printf "%-14s " $name
    timeout 300 python tools/time_hommul_batched.py 8 32 2>&1 | tail -1
  done
done
} > $OUT/${TAG}_mcs_ab.txt 2>&1
unset PHA_LIB_OVERRIDE
cat $OUT/${TAG}_mcs_ab.txt
echo "ab seconds: $(( $(date +%s) - T0 ))"

# per-kernel trace of one build (TRACE=name): the fused kernel's own time against the two it replaces
if [ -n "$TRACE" ]; then
  cd /tmp && export TMPDIR=/tmp
  for name in $TRACE; do
    export PHA_LIB_OVERRIDE=$LIBDIR/libphantom_amd_$name.so
    rm -rf /tmp/prof_mcs
    timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_mcs -o trace -- python $R/tools/traffic_probe.py hommul_batched:32 > $OUT/${TAG}_trace_$name.log 2>&1
    python - <<PY
import csv, glob, collections
rows=[]
for f in glob.glob("/tmp/prof_mcs/**/*kernel_trace.csv", recursive=True): rows += list(csv.DictReader(open(f)))
agg=collections.defaultdict(lambda:[0,0.0])
for r in rows:
    k=(r["Kernel_Name"].split("(")[0][:90], r["Grid_Size_X"], r["Grid_Size_Z"])
    agg[k][0]+=1; agg[k][1]+=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3
print("== $name: us per op (B = 32), kernels of the library")
for k,(c,t) in sorted(agg.items(), key=lambda kv:-kv[1][1]):
    if "pha" in k[0]: print(f"{t/c/32:8.2f} us/op  x{c:3d}  {k}")
PY
  done > $OUT/${TAG}_mcs_trace.txt 2>&1
  cat $OUT/${TAG}_mcs_trace.txt
fi
