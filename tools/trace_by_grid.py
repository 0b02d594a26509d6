"""Aggregate a rocprofv3 kernel_trace.csv by (kernel, grid size): mean/min GPU-side duration."""
import csv, glob, sys, collections
agg = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].replace("pha::", "")
        if "ntt_pass" not in name and len(sys.argv) < 3: continue
        name = name.replace("void ntt_pass_kernel", "ntt").replace("(NttKArgs)", "")
        g = (int(r["Grid_Size_X"]) // max(1,int(r["Workgroup_Size_X"])), int(r["Grid_Size_Y"]), int(r["Grid_Size_Z"]))
        agg[(name, g)].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for (name, g), v in sorted(agg.items()):
    v = sorted(v)[: max(1, len(v) - 2)]
    if "true, 0, false" not in name and "true, 1, false" not in name: continue
    print(f"{name:70s} grid={g}  n={len(v)}  mean={sum(v)/len(v)/1000:.2f}us  min={v[0]/1000:.2f}us")
