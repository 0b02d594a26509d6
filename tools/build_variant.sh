#!/bin/bash
# tools/build_variant.sh NAME "<extra -D flags>" [sources...]: an experiment build of the product library with compile-time switches
# (PHA_X_NT, PHA_X_SPLIT16, PHA_X_VARIANT, PHA_X_KNOBS, ...) -> phantom-fhe_amd/phantom_fhe_amd/libphantom_amd_NAME.so, loaded through
# PHA_LIB_OVERRIDE.  Only the listed sources (default: pha_ntt.hip) are recompiled; the rest links the product's objects.
set -e
NAME=$1; FLAGS=$2; shift 2 || true
SRCS=${@:-pha_ntt.hip}
cd "$(dirname "$0")/../phantom-fhe_amd/csrc"
mkdir -p var/$NAME
OBJS=""
for f in *.hip; do
  if echo " $SRCS " | grep -q " $f "; then
    /opt/rocm/bin/hipcc -O3 -Wall -Wno-unused-function -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off $FLAGS -c $f -o var/$NAME/${f%.hip}.o
    OBJS="$OBJS var/$NAME/${f%.hip}.o"
  else
    OBJS="$OBJS ${f%.hip}.o"
  fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../phantom_fhe_amd/libphantom_amd_$NAME.so $OBJS
echo "built libphantom_amd_$NAME.so"
