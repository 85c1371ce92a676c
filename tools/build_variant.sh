#!/bin/bash
# Build an experiment variant of libvpmi next to the product library (A/B inside one GPU session via VPMI_LIB).
# Usage: tools/build_variant.sh <name> [extra hipcc flags...]   ->  <pkg>/lib/libvpmi_<name>.so
set -e
NAME=$1; shift
PKG=$(dirname "$0")/../voiceprintrecognition-paddlepaddle_amd
OUT=$PKG/lib/libvpmi_$NAME.so
TMP=$(mktemp -d)
SRCS=$(python -c "import sys; sys.path.insert(0,'$PKG'); import build; print(' '.join(build.SOURCES))")
# the product flags (build.py): no packed-f32 instructions; WITH_PK=1 builds with them (the hazard study only)
NOPK="-Xclang -target-feature -Xclang -packed-fp32-ops"
[ -n "$WITH_PK" ] && NOPK=""
for s in $SRCS; do
  (/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $NOPK "$@" -c $PKG/csrc/$s -o $TMP/${s%.hip}.o 2>&1 | grep -v 'not a recognized feature' >&2) &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT $TMP/*.o
rm -rf $TMP
echo $OUT
