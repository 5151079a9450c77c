#!/bin/bash
# Round 3, GPU call 13: generic EASU with the lane's two consecutive rows filtered together (8-9 instead of 16 LDS reads per
# pixel) at 5 / 6 waves per SIMD against the tree; parity of the 6-wave build.
set -u
ROOT=$PWD; OUT=$ROOT/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
L=variants/libfsr1_
timeout 600 python tools/abtest.py --libs ${L}prev.so,${L}vp5.so,${L}vp6.so --workloads 1440p_to_4k,1270p_to_4k,1662p_to_4k,1440p_to_4k_x8,720p_to_1080p --kernels easu,pair --reps 3 > $OUT/r3c13_ab.log 2>&1
cat $OUT/r3c13_ab.log
FSR1_HIP_LIB=$ROOT/${L}vp6.so timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_special_values.py tests/test_gpu_bands.py tests/test_gpu_unorm.py tests/test_gpu_extents.py "tests/test_gpu_fullframe.py::test_whole_frame_two_pass_and_fused" -x -q -m gpu > $OUT/r3c13_pytest_vp6.log 2>&1; echo "rc=$?" >> $OUT/r3c13_pytest_vp6.log
tail -5 $OUT/r3c13_pytest_vp6.log
