#!/bin/bash
# usage: tools/build_variant.sh NAME "EXTRA -D flags"  -> variants/libfsr1_NAME.so (tuning builds for tools/abtest.py; not the product library)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; EXTRA=$2
B=/tmp/fsr1_variant_$NAME
rm -rf "$B"; mkdir -p "$B/pkg/csrc" "$B/include" "$ROOT/variants"
cp "$ROOT"/fidelityfx-fsr_amd/csrc/*.hip "$ROOT"/fidelityfx-fsr_amd/csrc/*.h "$ROOT"/fidelityfx-fsr_amd/csrc/*.c "$ROOT"/fidelityfx-fsr_amd/csrc/Makefile "$B/pkg/csrc/"
cp "$ROOT"/include/*.h "$ROOT"/include/*.hpp "$B/include/"
make -C "$B/pkg/csrc" -j8 EXTRA="$EXTRA" LIB="$ROOT/variants/libfsr1_$NAME.so" > "$B/build.log" 2>&1 || { tail -5 "$B/build.log"; exit 1; }
ls -la "$ROOT/variants/libfsr1_$NAME.so"
