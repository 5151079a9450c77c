#!/bin/bash
# Round 3, GPU call 11: row-pair form in both exact-2x kernels (EASU and fused): full test suite, A/B against the round-2 kernels.
set -u
ROOT=$PWD; OUT=$ROOT/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
L=variants/libfsr1_
timeout 1500 python -m pytest tests -x -q -m gpu --durations=3 > $OUT/r3c11_pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/r3c11_pytest.log
tail -8 $OUT/r3c11_pytest.log
timeout 900 python tools/abtest.py --libs ${L}r2base.so,@0 --workloads 1080p_to_4k,4k_to_8k_x16,540p_to_1080p,720p_to_1440p --kernels easu,pair,fused --reps 4 > $OUT/r3c11_ab.log 2>&1
cat $OUT/r3c11_ab.log
