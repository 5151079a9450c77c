#!/bin/bash
# Round 3, GPU call 1: the whole -m gpu suite on the new tree (self-launch bench, band flags, Hx2 / 10-bit pins, matrix-pipe
# EASU), then A/B of the matrix-pipe EASU kernel (same library, flag on / off) and of RCAS strip geometries, then the default bench line.
set -u
ROOT=$PWD; OUT=$ROOT/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -x -q -m gpu --durations=8 > $OUT/r3c1_pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/r3c1_pytest.log
tail -15 $OUT/r3c1_pytest.log
timeout 900 python tools/abtest.py --libs @0x1000,@0x800 --workloads 1080p_to_4k,4k_to_8k_x16,540p_to_1080p --kernels easu,pair --reps 3 > $OUT/r3c1_ab_mfma.log 2>&1
cat $OUT/r3c1_ab_mfma.log
L=variants/libfsr1_
timeout 900 python tools/abtest.py --libs @0,${L}rcas_r16w1.so,${L}rcas_r8w1.so,${L}rcas_r16.so --workloads 1080p_to_4k,1440p_to_4k_x8 --kernels rcas,rcas_cold,pair --reps 3 > $OUT/r3c1_ab_rcas.log 2>&1
cat $OUT/r3c1_ab_rcas.log
timeout 600 python bench.py > $OUT/r3c1_bench.json 2> $OUT/r3c1_bench.err; echo "bench rc=$?"
cut -c1-1500 $OUT/r3c1_bench.json
