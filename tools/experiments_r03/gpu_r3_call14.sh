#!/bin/bash
# Round 3, GPU call 14: exact-2x EASU with 12-byte tap reads (ds_read_b96 of R G B instead of ds_read_b128 of R G B luma):
# a power-bound kernel may trade LDS cycles for LDS bytes.
set -u
ROOT=$PWD; OUT=$ROOT/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
L=variants/libfsr1_
timeout 600 python tools/abtest.py --libs @0,${L}b96.so --workloads 1080p_to_4k,4k_to_8k_x16 --kernels easu,fused --reps 4 > $OUT/r3c14_ab.log 2>&1
cat $OUT/r3c14_ab.log
timeout 200 python tools/experiments_r03/power_probe.py --libs @0,${L}b96.so@0 --kernel easu --seconds 5 > $OUT/r3c14_power.log 2>&1; cat $OUT/r3c14_power.log
