#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_hb
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_hb -o trace -- python $R/tools/hommul_batched_probe.py 8 > /tmp/hb.log 2>&1
python - <<PY
import sys
sys.path.insert(0, "$R/tools")
import summarize_prof as S
S.by_grid("/tmp/prof_hb", "$R/gpurun_out/r04_hommul_batched_bygrid.csv")
tot=0
for l in open("$R/gpurun_out/r04_hommul_batched_bygrid.csv"):
    if l.startswith(("#","calls")) or "at::native" in l or "rocclr" in l: continue
    c,avg,mn,mx,grid,name=l.split(",",5)
    if int(c)>=5:
        tot+=float(avg)*int(c)/5
        print(f"{int(c)//5:2d}x {float(avg)/1e3:8.1f} us  {grid:14s} {name.strip()[:100]}")
print("sum per op-set of 8:", tot/1e3, "us ->", tot/8e3, "us per op")
PY
