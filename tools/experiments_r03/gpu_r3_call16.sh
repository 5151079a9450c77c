#!/bin/bash
# Round 3, GPU call 16: EXACT exact-2x EASU with the quad's analyses loaded once (70 VGPRs) against the tree; rcp_ieee now in RCAS / colour only.
set -u
ROOT=$PWD; OUT=$ROOT/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
L=variants/libfsr1_
timeout 600 python tools/abtest.py --libs ${L}prev.so,@0,${L}xshare.so --workloads 1080p_to_4k,4k_to_8k_x16 --kernels easu,rcas,pair --math exact --reps 3 > $OUT/r3c16_ab_exact.log 2>&1
cat $OUT/r3c16_ab_exact.log
