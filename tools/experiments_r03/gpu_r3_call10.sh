#!/bin/bash
# Round 3, GPU call 10: exact-2x EASU with the two pixels of a quad row filtered together (28 instead of 52 LDS reads per quad)
# at 5 / 7 / 8 waves per SIMD (85 / 68 / 64 VGPRs) against the tree; parity of the 7-wave build.
set -u
ROOT=$PWD; OUT=$ROOT/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
L=variants/libfsr1_
timeout 600 python tools/abtest.py --libs @0,${L}pair5.so,${L}pair7.so,${L}pair8.so --workloads 1080p_to_4k,4k_to_8k_x16,540p_to_1080p --kernels easu,pair --reps 3 > $OUT/r3c10_ab.log 2>&1
cat $OUT/r3c10_ab.log
FSR1_HIP_LIB=$ROOT/${L}pair7.so timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_special_values.py "tests/test_gpu_fullframe.py::test_whole_frame_two_pass_and_fused" -x -q -m gpu > $OUT/r3c10_pytest_pair7.log 2>&1; echo "rc=$?" >> $OUT/r3c10_pytest_pair7.log
tail -5 $OUT/r3c10_pytest_pair7.log
