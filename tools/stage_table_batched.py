"""Per-stage table of ONE HomMul + relinearize + rescale INSIDE A BATCH (the throughput half of BASELINE's metric) at the C3 set,
from a rocprofv3 kernel trace of `tools/traffic_probe.py hommul_batched:B` (5 op sets of B ciphertext pairs through
pha_tensor_prod_2x2_batched + pha_keyswitch_rescale_batched): mean GPU time per stage and per op (kernel time / B), algorithmic bytes
per op (SURVEY.md 8(d) per-unit figures x units; the key of the inner product is counted once per op set), fraction of the 8 TB/s line.
Several batch sizes go into one file -> profiles/stages_batched.json, which bench.py attaches to `hommul_relin_rescale.batched.stages`.
Usage: stage_table_batched.py <out.json> <B>=<trace dir> [<B>=<trace dir> ...]"""
import csv, glob, json, sys, time

N, QL, ALPHA, BETA = 1 << 16, 45, 15, 3
QLP = QL + ALPHA
W = 8 * N   # bytes of one limb


def stages_for(B, fused_conv, fused_resc=False):
    # (label, number of launches, name fragments any of which every launch must carry, algorithmic bytes PER OP)
    if fused_conv:   # r05: modup_conv_s1_kernel = the conversion as the load of the forward transform's strided pass; then the contiguous pass
        modup = [("mod-up: base conversion of 3 digits FUSED with the strided pass of the forward NTT of the converted limbs (r06: no own-limb copy, the inner product reads those limbs from c2)", 1,
                  ("modup_conv_s1",), BETA * (ALPHA + QL) * W + BETA * QL * W),
                 ("mod-up: forward NTT of the converted limbs, contiguous pass", 1, ("ntt_",), BETA * QL * W)]
    else:
        modup = [("mod-up: base conversion, 3 digits (+ verbatim copy of each digit's own limbs)", 1, ("bconv_kernel",), BETA * (ALPHA + QL) * W),
                 ("mod-up: forward NTT of the converted limbs (strided pass, contiguous pass)", 2, ("ntt_",), 2 * BETA * QL * W)]
    return [
        ("tensor product (multiply), B ciphertext pairs in one launch", 1, ("ew_kernel", "tensor"), 7 * QL * W),
        ("mod-up: inverse NTT x partQlHatInv (contiguous pass, strided pass)", 2, ("ntt_",), 2 * QL * W),
    ] + modup + [
        ("key inner product (key limbs in registers across the batch)", 1, ("inner_prod",), QLP * (BETA + 2) * W + QLP * 2 * BETA * W // B),
        ("mod-down + rescale: inverse NTT of P and last limb, 2 polys (contiguous pass, strided pass)", 2, ("ntt_",), 2 * 2 * (ALPHA + 1) * W),
    ] + ([   # r06: the conversion as the load of the final forward transform's strided pass (rescale form of modup_conv_s1_kernel)
        ("mod-down + rescale: conversion + last-limb fold FUSED with the strided pass of the ONE forward NTT", 1, ("modup_conv_s1",),
         2 * (ALPHA + 1 + QL - 1) * W + 2 * (QL - 1) * W),
        ("mod-down + rescale: contiguous pass of that forward NTT, epilogue (ct + cx/P - .)/q_last", 1, ("ntt_",), 2 * (QL - 1) * (1 + 2) * W),
    ] if fused_resc else [
        ("mod-down + rescale: conversion + last-limb fold", 1, ("bconv_rescale_kernel",), 2 * (ALPHA + 1 + QL - 1) * W),
        ("mod-down + rescale: ONE forward NTT, epilogue (ct + cx/P - .)/q_last (strided pass, contiguous pass)", 2, ("ntt_",), 2 * (QL - 1) * (2 + 2) * W),
    ])


def table_for(B, trace_dir):
    rows = []
    for f in glob.glob(trace_dir + "/**/*kernel_trace.csv", recursive=True):
        rows += list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    lib = [r for r in rows if any(k in r["Kernel_Name"] for k in ("ntt_", "bconv", "inner_prod", "ew_kernel", "modup_ip", "modup_conv"))]
    fused_conv = any("modup_conv_s1" in r["Kernel_Name"] for r in lib)
    st = stages_for(B, fused_conv, fused_conv and not any("bconv_rescale_kernel" in r["Kernel_Name"] for r in lib))
    per_set = sum(cnt for _, cnt, _, _ in st)
    sets = len(lib) // per_set
    assert sets >= 2 and len(lib) % per_set == 0, f"{len(lib)} library kernels in the trace, {per_set} per op set expected: " + \
        " | ".join(r["Kernel_Name"][:40] for r in lib[:per_set + 2])
    lib = lib[-(sets - 1) * per_set:]    # drop the first op set (cold)
    sets -= 1
    table, pos = [], 0
    for label, cnt, frags, nbytes in st:
        us, grids, each, names = 0.0, [], [], []
        for j in range(cnt):
            ks = [lib[o * per_set + pos + j] for o in range(sets)]
            assert all(any(fr in k["Kernel_Name"] for fr in frags) for k in ks), (label, frags, ks[0]["Kernel_Name"])
            one = sum(int(k["End_Timestamp"]) - int(k["Start_Timestamp"]) for k in ks) / sets / 1e3
            each.append(round(one / B, 2))
            us += one
            k0 = ks[0]
            grids.append(f'{int(k0["Grid_Size_X"]) // int(k0["Workgroup_Size_X"])}x{k0["Grid_Size_Y"]}x{k0["Grid_Size_Z"]}')
            names.append(k0["Kernel_Name"].split("(")[0].replace("void pha::", "")[:70])
        pos += cnt
        table.append({"stage": label, "kernels": names, "grids": grids, "us_per_op": round(us / B, 2), "kernels_us_per_op": each,
                      "algorithmic_bytes_per_op": nbytes, "frac_of_8TBps": round(nbytes / (us / B * 1e-6) / 8e12, 4)})
    span = [(int(lib[o * per_set]["Start_Timestamp"]), int(lib[o * per_set + per_set - 1]["End_Timestamp"])) for o in range(sets)]
    worst = min(table, key=lambda t: t["frac_of_8TBps"])
    return {"batch": B, "per_op_us_sum_of_kernels": round(sum(t["us_per_op"] for t in table), 2),
            "per_op_us_first_start_to_last_end": round(sum(b - a for a, b in span) / sets / B / 1e3, 2),
            "algorithmic_bytes_per_op": sum(t["algorithmic_bytes_per_op"] for t in table),
            "furthest_below_roofline": worst["stage"], "furthest_below_roofline_frac": worst["frac_of_8TBps"],
            "stages": table, "op_sets_averaged": sets}


out = sys.argv[1]
doc = {"source": "rocprofv3 --kernel-trace of tools/traffic_probe.py hommul_batched:B (tools/profile_r06.sh); under the profiler every launch "
                 "is serialised, so the per-op sum is a few per cent above the figure bench.py times without it",
       "collected": time.strftime("%Y-%m-%dT%H:%M:%SZ", time.gmtime()), "batches": []}
for arg in sys.argv[2:]:
    b, d = arg.split("=", 1)
    t = table_for(int(b), d)
    doc["batches"].append(t)
    print(f'--- B = {t["batch"]}: {t["per_op_us_sum_of_kernels"]} us per op (sum of kernels), span {t["per_op_us_first_start_to_last_end"]}')
    for s in t["stages"]:
        print(f'{s["us_per_op"]:8.2f} us  {s["frac_of_8TBps"]:.3f}  {s["stage"]}  {s["grids"]}  {s["kernels_us_per_op"]}')
    print("furthest below the roofline:", t["furthest_below_roofline"], t["furthest_below_roofline_frac"])
best = min(doc["batches"], key=lambda t: t["per_op_us_sum_of_kernels"])
doc["furthest_below_roofline"] = best["furthest_below_roofline"]
doc["best_batch"] = best["batch"]
json.dump(doc, open(out, "w"), indent=1)
