#!/bin/bash
# Round 3, GPU call 6: why is the generic fused kernel slower at 1.3x than in round 2?  Variants without the lane permutation /
# without the bounds-from-taps in its EASU phase.
set -u
ROOT=$PWD; OUT=$ROOT/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
L=variants/libfsr1_
timeout 900 python tools/abtest.py --libs ${L}r2base.so,@0,${L}fg_noperm.so,${L}fg_notapb.so,${L}fg_neither.so --workloads 831p_to_1080p,1662p_to_4k,720p_to_1080p,1440p_to_4k --kernels fused --reps 2 > $OUT/r3c6_ab.log 2>&1
cat $OUT/r3c6_ab.log
