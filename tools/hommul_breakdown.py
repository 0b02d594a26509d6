"""Per-kernel GPU time of ONE HomMul + relinearize + rescale, from a rocprofv3 kernel_trace.csv of bench.py
(runs on the GPU box).  Kernels of the HomMul leg are those launched with a grid that occurs `steps` times."""
import csv, glob, sys, collections
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
agg = collections.OrderedDict()
for r in rows:
    name = r["Kernel_Name"].replace("pha::", "").replace("void ", "")
    name = name.replace("ntt_pass_kernel", "ntt").replace("(NttKArgs)", "").replace("(BConvLaunch)", "")
    g = (int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])), int(r["Grid_Size_Y"]), int(r["Grid_Size_Z"]))
    agg.setdefault((name, g), []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
total = 0.0
print(f"# kernels launched a multiple of {steps} times (the HomMul leg), in first-launch order")
for (name, g), v in agg.items():
    if len(v) % steps or len(v) > 4 * steps:
        continue
    per_op = len(v) // steps
    mean = sum(v) / len(v) / 1000
    total += mean * per_op
    print(f"{mean * per_op:8.2f} us/op  ({per_op} x {mean:7.2f})  grid={g}  {name[:110]}")
print(f"{total:8.2f} us/op  sum of kernel times")
