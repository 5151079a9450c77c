#!/bin/bash
# TEST INFRASTRUCTURE.  Builds oracle/_ref/libfsr1_ref.so = the reference's own ffx_a.h + ffx_fsr1.h
# compiled from where they lie under /root/reference (never copied into the repo):
#   * ref_con.c   : the A_CPU build (constant setup), plain C, headers included in place;
#   * ref_wrap.cpp: the A_GPU/A_GLSL build (FsrEasuF/H, FsrRcasF/H) through ref_glsl_shim.hpp.  GLSL's
#     `out`/`inout` parameter qualifiers have no C++ spelling, so a sed pass writes a *temporary*
#     copy of the two headers with `out T x`/`inout T x` -> `T& x` (and the in*/out*/inout* macro
#     family of ffx_a.h:2435-2472 mapped likewise); the temp dir is deleted after compiling.
# Pinned semantics: -ffp-contract=off, IEEE minNum/maxNum with -0 < +0, correctly rounded 1/x (see shim header).
# When /root/reference is absent (the GPU box) the prebuilt .so that travelled with the tree is kept.
set -euo pipefail
HERE=$(cd "$(dirname "$0")" && pwd)
REF=${FSR_REFERENCE:-/root/reference}/ffx-fsr
OUT=$HERE/_ref
if [ ! -f "$REF/ffx_fsr1.h" ]; then
  if [ -f "$OUT/libfsr1_ref.so" ]; then echo "build_ref: reference absent, keeping prebuilt $OUT/libfsr1_ref.so"; exit 0; fi
  echo "build_ref: reference absent and no prebuilt library" >&2; exit 3
fi
mkdir -p "$OUT"
TMP=$(mktemp -d); trap 'rm -rf "$TMP"' EXIT
rewrite() {
  sed -E \
    -e 's/^([[:space:]]*#define[[:space:]]+in(A[A-Z]{1,2}[1-4]))[[:space:]]+in[[:space:]]+(A[A-Z]{1,2}[1-4])[[:space:]]*$/\1 \3/' \
    -e 's/^([[:space:]]*#define[[:space:]]+(out|inout)(A[A-Z]{1,2}[1-4]))[[:space:]]+(out|inout)[[:space:]]+(A[A-Z]{1,2}[1-4])[[:space:]]*$/\1 \5\&/' \
    -e 's/\b(inout|out)[[:space:]]+(A[A-Z]{1,2}[1-4])[[:space:]]+([A-Za-z_][A-Za-z0-9_]*)/\2\& \3/g' \
    "$1" > "$2"
}
rewrite "$REF/ffx_a.h" "$TMP/ref_ffx_a.h"
rewrite "$REF/ffx_fsr1.h" "$TMP/ref_ffx_fsr1.h"
CXXFLAGS="-std=c++17 -O2 -ffp-contract=off -fno-fast-math -fpermissive -fopenmp -fPIC -w"
g++ $CXXFLAGS -I"$TMP" -I"$HERE" -c "$HERE/ref_wrap.cpp" -o "$TMP/ref_wrap.o"
gcc -O2 -ffp-contract=off -fPIC -I"$REF" -c "$HERE/ref_con.c" -o "$TMP/ref_con.o"
g++ -shared -fopenmp -o "$OUT/libfsr1_ref.so" "$TMP/ref_wrap.o" "$TMP/ref_con.o" -lm
echo "build_ref: built $OUT/libfsr1_ref.so"
# A SECOND compilation of the same sources with multiply-adds CONTRACTED (-ffp-contract=fast -mfma): what a shading-language compiler
# is free to do to the reference's `a*b+c` expressions unless they are marked `precise`.  Never a parity target — it exists only so that
# tests/ref_self_spread.py can publish how far two conformant compilations of the reference lie from each other (README.md), the
# yardstick for the product's own distance from the pinned build.  Built only where the host CPU has FMA; constant setup stays uncontracted.
if grep -qw fma /proc/cpuinfo; then
  g++ ${CXXFLAGS/-ffp-contract=off/-ffp-contract=fast} -mfma -I"$TMP" -I"$HERE" -c "$HERE/ref_wrap.cpp" -o "$TMP/ref_wrap_fma.o"
  g++ -shared -fopenmp -o "$OUT/libfsr1_ref_fma.so" "$TMP/ref_wrap_fma.o" "$TMP/ref_con.o" -lm
  echo "build_ref: built $OUT/libfsr1_ref_fma.so (contracted twin, self-spread only)"
fi
