#!/bin/bash
# Round 3, GPU call 5: band form of the fused quad kernel + row terms one row ahead: band / parity tests, the auto rule at
# launch-bound sizes (two dispatches vs fused, generic kernels), generic kernels once more against round 2.
set -u
ROOT=$PWD; OUT=$ROOT/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
L=variants/libfsr1_
timeout 900 python -m pytest tests/test_gpu_bands.py tests/test_gpu_parity.py tests/test_runner.py tests/test_gpu_unorm.py tests/test_gpu_extents.py -x -q -m gpu > $OUT/r3c5_pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/r3c5_pytest.log
tail -6 $OUT/r3c5_pytest.log
timeout 900 python tools/abtest.py --libs ${L}r2base.so,@0 --workloads 720p_to_1080p,831p_to_1080p,1440p_to_4k,1662p_to_4k,1440p_to_4k_x8 --kernels easu,pair,fused --reps 3 > $OUT/r3c5_ab.log 2>&1
cat $OUT/r3c5_ab.log
timeout 300 runner/fsr1_runner --gpus 1 --bands --pipeline fused --steps 300 --warmup 30 > $OUT/r3c5_runner_bands.log 2>&1; cat $OUT/r3c5_runner_bands.log
timeout 300 runner/fsr1_runner --gpus 1 --bands --steps 300 --warmup 30 >> $OUT/r3c5_runner_bands.log 2>&1; tail -1 $OUT/r3c5_runner_bands.log
