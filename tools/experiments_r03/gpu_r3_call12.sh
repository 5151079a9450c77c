#!/bin/bash
# Round 3, GPU call 12: phase 2 reading its five lumas as whole records (conflict-free ds_read_b128) against the tree before.
set -u
ROOT=$PWD; OUT=$ROOT/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
L=variants/libfsr1_
timeout 600 python tools/abtest.py --libs ${L}prev.so,@0 --workloads 1080p_to_4k,4k_to_8k_x16,1440p_to_4k,540p_to_1080p --kernels easu,fused --reps 4 > $OUT/r3c12_ab.log 2>&1
cat $OUT/r3c12_ab.log
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_special_values.py -x -q -m gpu > $OUT/r3c12_pytest.log 2>&1; echo "rc=$?" >> $OUT/r3c12_pytest.log; tail -3 $OUT/r3c12_pytest.log
