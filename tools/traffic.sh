#!/bin/bash
# HBM-side bytes per forward-NTT step and per HomMul op: rocprofv3 --pmc FETCH_SIZE and WRITE_SIZE in SEPARATE runs (the
# MI355X guide's recipe), summed over the dispatches of tools/traffic_probe.py, written to gpurun_out/traffic.json
# (copy to profiles/traffic.json, which bench.py reports with its sha).
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
HB=${HB:-8}    # batch of the batched HomMul pass
for mode in ntt hommul hommul_batched; do
  arg=$mode; [ $mode = hommul_batched ] && arg=hommul_batched:$HB
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --pmc $c --output-format csv -d $OUT/tr_${mode}_$c -o pmc -- python $R/tools/traffic_probe.py $arg > $OUT/tr_${mode}_$c.log 2>&1
  done
done
export HB
python - <<'PY'
import csv, glob, json, os, collections, time
out = os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out"
NTT_STEPS, HM_OPS = 6, 6
HB = int(os.environ.get("HB", "8"))
HB_SETS = 5
res = {}
detail = {}
for mode, div in (("ntt", NTT_STEPS), ("hommul", HM_OPS), ("hommul_batched", HB_SETS * HB)):
    tot = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        s = 0.0
        per = collections.defaultdict(float)
        for f in glob.glob(f"{out}/tr_{mode}_{c}/**/*counter_collection*.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                name = r["Kernel_Name"]
                if not any(k in name for k in ("pha::", "pha_", "ntt_", "bconv", "inner_prod", "ew_kernel", "modup_ip")):
                    continue      # torch's RNG / copy kernels of the setup are not part of the op (the op's own copy_ is counted below)
                s += float(r["Counter_Value"])
                per[name.split("(")[0][:60]] += float(r["Counter_Value"])
        tot[c] = s
        detail[f"{mode}_{c}_KiB_per_unit"] = {k: v / div for k, v in sorted(per.items(), key=lambda kv: -kv[1])[:12]}
    # FETCH_SIZE / WRITE_SIZE count in KiB (MI355X_MICROARCH.md: 1 unit = 1 KiB); gfx950 correction: FETCH_SIZE x 2
    res[mode] = (2.0 * tot["FETCH_SIZE"] + tot["WRITE_SIZE"]) * 1024.0 / div
doc = {"ntt_batched_bytes_per_launch": res["ntt"], "hommul_bytes_per_op": res["hommul"],
       "hommul_batched_bytes_per_op": res["hommul_batched"], "hommul_batched_batch": HB,
       "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate runs of tools/traffic_probe.py (6 steps / 6 ops / 5 op sets of HB ciphertext pairs), summed over "
                 "the library's kernels, (2 x FETCH_SIZE + WRITE_SIZE) KiB per the gfx950 correction of MI355X_MICROARCH.md",
       "collected": time.strftime("%Y-%m-%dT%H:%M:%SZ", time.gmtime()), "detail": detail}
json.dump(doc, open(out + "/traffic.json", "w"), indent=1)
print(json.dumps({k: v for k, v in doc.items() if k != "detail"}, indent=1))
PY
rm -rf $OUT/tr_*/
