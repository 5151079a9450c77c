#!/bin/bash
# usage: tools/build_variant.sh NAME "EXTRA -D flags" [PATCH ...]  -> variants/libfsr1_NAME.so (tuning builds for tools/abtest.py; not the product library)
# PATCH: unified diffs (paths a/fidelityfx-fsr_amd/csrc/..., a/include/...) applied to the copy of the tree the variant is built from
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; EXTRA=$2; shift; shift || true
B=/tmp/fsr1_variant_$NAME
rm -rf "$B"; mkdir -p "$B/pkg/csrc" "$B/include" "$ROOT/variants"
cp "$ROOT"/fidelityfx-fsr_amd/csrc/*.hip "$ROOT"/fidelityfx-fsr_amd/csrc/*.h "$ROOT"/fidelityfx-fsr_amd/csrc/*.c "$ROOT"/fidelityfx-fsr_amd/csrc/*.cpp "$ROOT"/fidelityfx-fsr_amd/csrc/Makefile "$B/pkg/csrc/"
cp "$ROOT"/include/*.h "$ROOT"/include/*.hpp "$B/include/"
for P in "$@"; do (cd "$B" && sed -e "s#fidelityfx-fsr_amd/csrc/#pkg/csrc/#g" "$(cd "$ROOT" && realpath "$P")" | patch -p1 -s) || { echo "patch $P failed"; exit 1; }; done
make -C "$B/pkg/csrc" -j8 EXTRA="$EXTRA" LIB="$ROOT/variants/libfsr1_$NAME.so" > "$B/build.log" 2>&1 || { tail -5 "$B/build.log"; exit 1; }
ls -la "$ROOT/variants/libfsr1_$NAME.so"
