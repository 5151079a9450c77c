#!/bin/bash
# Round 3, GPU call 7: row terms in LDS — per kernel: generic EASU with / without, generic fused with / without (same box).
set -u
ROOT=$PWD; OUT=$ROOT/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
L=variants/libfsr1_
timeout 600 python -m pytest tests/test_gpu_bands.py tests/test_gpu_parity.py -x -q -m gpu > $OUT/r3c7_pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/r3c7_pytest.log
tail -4 $OUT/r3c7_pytest.log
timeout 900 python tools/abtest.py --libs ${L}r2base.so,@0,${L}easu_norowt.so,${L}fused_rowt.so --workloads 831p_to_1080p,1662p_to_4k,720p_to_1080p,1440p_to_4k,1270p_to_4k,1440p_to_4k_x8 --kernels easu,fused --reps 3 > $OUT/r3c7_ab.log 2>&1
cat $OUT/r3c7_ab.log
