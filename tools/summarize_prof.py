"""Condense rocprofv3 output directories into small text summaries (runs on the GPU box)."""
import csv, glob, os, sys, collections

def short(name):
    name = name.replace("pha::", "")
    return name if len(name) < 150 else name[:147] + "..."

def stats(d, out):
    for f in glob.glob(os.path.join(d, "**", "*kernel_stats*.csv"), recursive=True):
        with open(f) as fh, open(out, "w") as o:
            rows = list(csv.DictReader(fh))
            o.write(f"# rocprofv3 --kernel-trace --stats summary ({os.path.basename(f)})\n")
            o.write("calls,total_ns,avg_ns,min_ns,max_ns,pct,name\n")
            for r in rows:
                o.write(f"{r.get('Calls')},{r.get('TotalDurationNs')},{r.get('AverageNs')},{r.get('MinNs')},{r.get('MaxNs')},{r.get('Percentage')},\"{short(r.get('Name',''))}\"\n")
        return True
    return False

def by_grid(d, out):
    """kernel_trace.csv -> calls / mean / min duration per (kernel, grid): the stats file lumps the grids of one
    template instantiation together, the bench's launch pair is the (16,45,1) + (128,45,1) rows here."""
    files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    if not files:
        return False
    agg = collections.OrderedDict()
    rows = []
    for f in files:
        with open(f) as fh:
            rows += list(csv.DictReader(fh))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    for r in rows:
        g = (int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])), int(r["Grid_Size_Y"]), int(r["Grid_Size_Z"]))
        agg.setdefault((short(r["Kernel_Name"]), g), []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    with open(out, "w") as o:
        o.write("# rocprofv3 --kernel-trace: GPU-side duration per (kernel, workgroups x, y, z), first-launch order\n")
        o.write("calls,avg_ns,min_ns,max_ns,grid,name\n")
        for (name, g), v in agg.items():
            o.write(f"{len(v)},{sum(v) / len(v):.0f},{min(v)},{max(v)},{g[0]}x{g[1]}x{g[2]},\"{name}\"\n")
    return True

def pmc(d, out):
    files = glob.glob(os.path.join(d, "**", "*counter_collection*.csv"), recursive=True)
    if not files:
        return False
    agg = collections.defaultdict(lambda: [0.0, 0])
    meta = {}
    for f in files:
        with open(f) as fh:
            for r in csv.DictReader(fh):
                k = (short(r["Kernel_Name"]) + " grid=" + str(int(r["Grid_Size"]) // max(1, int(r["Workgroup_Size"]))), r["Counter_Name"])
                agg[k][0] += float(r["Counter_Value"]); agg[k][1] += 1
                meta[k[0]] = (r.get("Grid_Size"), r.get("Workgroup_Size"), r.get("LDS_Block_Size"), r.get("VGPR_Count"), r.get("SGPR_Count"))
    with open(out, "w") as o:
        o.write("# rocprofv3 --pmc per-kernel mean per dispatch\nkernel,grid,wg,lds,vgpr,sgpr,counter,mean,dispatches\n")
        for (k, c), (s, n) in sorted(agg.items()):
            m = meta[k]
            o.write(f"\"{k}\",{m[0]},{m[1]},{m[2]},{m[3]},{m[4]},{c},{s/n:.1f},{n}\n")
    return True

if __name__ == "__main__":
    base = sys.argv[1]
    tag = sys.argv[2] if len(sys.argv) > 2 else "r01"
    stats(os.path.join(base, "prof_trace"), os.path.join(base, f"{tag}_kernel_stats.csv"))
    by_grid(os.path.join(base, "prof_trace"), os.path.join(base, f"{tag}_kernel_by_grid.csv"))
    for name in ("sq", "lds", "fetch", "write"):
        pmc(os.path.join(base, f"prof_pmc_{name}"), os.path.join(base, f"{tag}_pmc_{name}.csv"))
