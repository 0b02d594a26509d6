#!/bin/bash
# kernel timeline (start / end relative to the op's first kernel, queue) of the LAST single HomMul + relinearize + rescale of the probe
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_hs
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_hs -o trace -- python $R/tools/hommul_single_probe.py 6 > /tmp/hs.log 2>&1
python - <<PY
import csv, glob
f = glob.glob("/tmp/prof_hs/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "at::native" not in r["Kernel_Name"] and "rocclr" not in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last op starts at the last ew_kernel
last = max(i for i, r in enumerate(rows) if "ew_kernel" in r["Kernel_Name"])
t0 = int(rows[last]["Start_Timestamp"])
for r in rows[last:]:
    print(f'{(int(r["Start_Timestamp"]) - t0) / 1e3:8.1f} -> {(int(r["End_Timestamp"]) - t0) / 1e3:8.1f} us  q{r["Queue_Id"]:>3s}  {r["Grid_Size_X"]:>7s}x{r["Grid_Size_Y"]:>5s}x{r["Grid_Size_Z"]:>3s}  {r["Kernel_Name"][:90]}')
PY
