#!/bin/bash
# Round 3, GPU call 8: exact-2x EASU with the quad's analyses loaded once and its bounds taken of the first pixel's taps (52
# instead of 60 LDS reads per quad, 57 instead of 48 VGPRs) against the tree.
set -u
ROOT=$PWD; OUT=$ROOT/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
L=variants/libfsr1_
timeout 600 python tools/abtest.py --libs @0,${L}s2share.so --workloads 1080p_to_4k,4k_to_8k_x16,540p_to_1080p --kernels easu,pair --reps 4 > $OUT/r3c8_ab.log 2>&1
cat $OUT/r3c8_ab.log
