#!/bin/bash
# tools/r06_final.sh [TAG]: the round's final evidence set from a PROVABLY fresh build.
#   1. here (the build container, no GPU): `make clean && make` of the HIP libraries and the oracle, so the .so files that travel to the
#      GPU box are built from the tree as it stands; their sha256 and `git rev-parse HEAD` / `git status --short` go to
#      profiles/<TAG>_build.txt (hipcc cross-compiles gfx950; compiling on the GPU box would burn ~3 GPU-minutes);
#   2. on a fresh MI355X box (gpurun): tools/r06_gpu.sh -- checks those sha256 against the files it loads, full `pytest -m gpu`,
#      smoke(), tools/profile_r06.sh (bench line, kernel traces, stage tables single + batched, PMC passes, traffic), the reference's
#      benchmark shapes (ntt_sweep / keyswitch_sweep / ckks_ops_bench);
#   3. here again: tools/collect_profiles.sh copies the tagged summaries from gpurun_out/ into profiles/.
set -e
TAG=${1:-r06z}
cd "$(dirname "$0")/.."
ROOT=$(pwd)
make -C phantom-fhe_amd/csrc clean > /dev/null
rm -f oracle/liboracle.so oracle/liboracle_native.so
rm -f phantom-fhe_amd/phantom_fhe_amd/libphantom_amd_*.so        # experiment builds of earlier sessions (tools/build_variant.sh)
rm -rf phantom-fhe_amd/csrc/var
python -c "import __graft_entry__ as g; g.build()" > /tmp/r06_build.log 2>&1 || { tail -30 /tmp/r06_build.log; exit 1; }
DIRTY=$(git status --short | grep -v '^??' | grep -v "profiles/${TAG}_" || true)      # (the evidence files of this tag are rewritten by this very run)
{
  echo "# built by tools/r06_final.sh in the build container (hipcc --offload-arch=gfx950), $(date -u +%Y-%m-%dT%H:%M:%SZ)"
  echo "HEAD $(git rev-parse HEAD)"
  echo "dirty-files $(printf '%s' "$DIRTY" | grep -c . || true)"
  printf '%s\n' "$DIRTY" | sed '/^$/d; s/^/  /'
  sha256sum phantom-fhe_amd/phantom_fhe_amd/*.so oracle/*.so
} > profiles/${TAG}_build.txt
cat profiles/${TAG}_build.txt
/usr/local/graft/bin/gpurun --timeout 2400 -- "TAG=$TAG bash tools/r06_gpu.sh" 2>&1 | tail -60
bash tools/collect_profiles.sh $TAG
