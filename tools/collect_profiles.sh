#!/bin/bash
# tools/collect_profiles.sh TAG: copy the tagged summaries of a tools/r06_final.sh run from gpurun_out/ (scratch) into profiles/
# (tracked), plus stages.json / traffic.json, which bench.py reports with their sha.
TAG=${1:?usage: collect_profiles.sh TAG}
cd "$(dirname "$0")/.."
for f in gpurun_out/${TAG}_*; do
  case "$f" in *pytest.txt|*.err) continue ;; esac
  cp "$f" profiles/
done
cp gpurun_out/stages.json gpurun_out/stages_batched.json gpurun_out/traffic.json profiles/ 2>/dev/null
ls profiles | grep "^${TAG}_"
