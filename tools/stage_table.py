"""Per-stage table of ONE HomMul + relinearize + rescale at the C3 set from a rocprofv3 kernel trace of
`tools/traffic_probe.py hommul` (6 ops): mean GPU time per stage, algorithmic bytes (SURVEY.md 8(d) per-unit figures x units),
fraction of the 8 TB/s line -> profiles/stages.json, which bench.py attaches to `hommul_relin_rescale.stages` so that the kernel
furthest below the roofline is named in the bench line.  Usage: stage_table.py <trace dir> <out.json>"""
import csv, glob, json, sys, time

N, QL, ALPHA, BETA = 1 << 16, 45, 15, 3
QLP = QL + ALPHA
W = 8 * N   # bytes of one limb
# (label, kernel-name fragments in launch order, algorithmic bytes)
STAGES = [
    ("tensor product (multiply)", ["ew_kernel<6>"], 7 * QL * W),                                   # 4 reads + 3 writes per limb
    ("mod-up: inverse NTT x partQlHatInv", ["ntt_pass_kernel", "ntt_pass_kernel"], 2 * QL * W),
    ("mod-up: base conversion, 3 digits", ["bconv_kernel"], BETA * (ALPHA + QL) * W),             # in 15 + out 30 + 15 per digit
    # r03: the inner product is the epilogue of the forward transform's contiguous pass (modup_ip_kernel); algorithmic bytes of both
    # r04: the fused kernel also runs the contiguous pass of the mod-down's inverse transform on the special limbs and the last data
    # limb (ModupIpArgs::inv_from), so that inverse is ONE launch (the strided pass); its algorithmic bytes are split evenly
    ("mod-up: forward NTT of the converted limbs + key inner product (fused; + contiguous pass of the inverse of P and last limb)",
     ["ntt_pass_kernel", "modup_ip_kernel"], 2 * BETA * QL * W + QLP * (3 * BETA + 2) * W + 2 * (ALPHA + 1) * W),
    ("mod-down + rescale: inverse NTT of P and last limb, 2 polys: strided pass", ["ntt_pass_kernel"], 2 * (ALPHA + 1) * W),
    ("mod-down + rescale: conversion + last-limb fold", ["bconv_rescale_kernel"], 2 * (ALPHA + 1 + QL - 1) * W),
    ("mod-down + rescale: ONE forward NTT, epilogue (ct + cx/P - .)/q_last", ["ntt_pass_kernel", "ntt_pass_kernel"], 2 * (QL - 1) * (2 + 2) * W),
]
PER_OP = sum(len(k) for _, k, _ in STAGES)

rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
lib = [r for r in rows if any(k in r["Kernel_Name"] for k in ("ntt_pass_kernel", "bconv", "inner_prod", "ew_kernel", "modup_ip"))]
ops = len(lib) // PER_OP
assert ops >= 2, f"{len(lib)} library kernels in the trace, {PER_OP} per op expected"
lib = lib[-(ops - 1) * PER_OP:]          # drop the first op (cold)
ops -= 1
table, pos = [], 0
for label, frags, nbytes in STAGES:
    us, grids, each = 0.0, [], []
    for j, frag in enumerate(frags):
        ks = [lib[o * PER_OP + pos + j] for o in range(ops)]
        assert all(frag in k["Kernel_Name"] for k in ks), (label, frag, ks[0]["Kernel_Name"])
        one = sum(int(k["End_Timestamp"]) - int(k["Start_Timestamp"]) for k in ks) / ops / 1e3
        each.append(round(one, 2))
        us += one
        k0 = ks[0]
        grids.append(f'{int(k0["Grid_Size_X"]) // int(k0["Workgroup_Size_X"])}x{k0["Grid_Size_Y"]}x{k0["Grid_Size_Z"]}')
    pos += len(frags)
    table.append({"stage": label, "grids": grids, "us": round(us, 2), "kernels_us": each, "algorithmic_bytes": nbytes,
                  "frac_of_8TBps": round(nbytes / (us * 1e-6) / 8e12, 4)})
span = [(int(lib[o * PER_OP]["Start_Timestamp"]), int(lib[o * PER_OP + PER_OP - 1]["End_Timestamp"])) for o in range(ops)]
worst = min(table, key=lambda t: t["frac_of_8TBps"])
doc = {"per_op_us_sum_of_kernels": round(sum(t["us"] for t in table), 2),
       "per_op_us_first_start_to_last_end": round(sum(b - a for a, b in span) / ops / 1e3, 2),
       "furthest_below_roofline": worst["stage"], "stages": table, "ops_averaged": ops,
       "source": "rocprofv3 --kernel-trace of tools/traffic_probe.py hommul (tools/profile_r04.sh)",
       "collected": time.strftime("%Y-%m-%dT%H:%M:%SZ", time.gmtime())}
json.dump(doc, open(sys.argv[2], "w"), indent=1)
for t in table:
    print(f'{t["us"]:8.2f} us  {t["frac_of_8TBps"]:.3f}  {t["stage"]}  {t["grids"]}  {t["kernels_us"]}')
print("sum", doc["per_op_us_sum_of_kernels"], "span", doc["per_op_us_first_start_to_last_end"], "worst:", worst["stage"])
