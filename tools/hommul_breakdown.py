"""Mean GPU time per (kernel, grid) from a rocprofv3 kernel_trace.csv of bench.py, in first-launch order
(runs on the GPU box).  The HomMul + relinearize + rescale leg is the run of rows after the dyadic tensor kernel."""
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
for (name, g), v in agg.items():
    if "at::" in name or "elementwise" in name or len(v) < 4:
        continue
    v = sorted(v)
    print(f"{sum(v) / len(v) / 1000:8.2f} us mean {v[0] / 1000:8.2f} min  n={len(v):4d}  grid={g}  {name[:100]}")
