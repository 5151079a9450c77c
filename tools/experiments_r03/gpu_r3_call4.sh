#!/bin/bash
# Round 3, GPU call 4: the tree without the luma plane (exact-2x kernels byte-identical to round 2 again) and with the lane ->
# column permutation in the packed-fp16 generic kernels too: tests, A/B against the round-2 kernels for F and H arithmetic.
set -u
ROOT=$PWD; OUT=$ROOT/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
L=variants/libfsr1_
timeout 1500 python -m pytest tests -x -q -m gpu --durations=5 > $OUT/r3c4_pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/r3c4_pytest.log
tail -12 $OUT/r3c4_pytest.log
timeout 900 python tools/abtest.py --libs ${L}r2base.so,@0 --workloads 1440p_to_4k,1270p_to_4k,1662p_to_4k,1440p_to_4k_x8,1080p_to_4k --kernels easu,pair,fused --reps 3 > $OUT/r3c4_ab_generic.log 2>&1
cat $OUT/r3c4_ab_generic.log
timeout 900 python tools/abtest.py --libs ${L}r2base.so,@0 --workloads 1440p_to_4k,1662p_to_4k,1440p_to_4k_x8 --kernels easu,pair,fused --math h --reps 2 > $OUT/r3c4_ab_generic_h.log 2>&1
cat $OUT/r3c4_ab_generic_h.log
