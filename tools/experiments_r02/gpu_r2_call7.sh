#!/bin/bash
# round-2 GPU call 7: RCAS ring refilled right after conversion (R rows ahead with R slots) vs the old ring; parity first
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_unorm.py tests/test_gpu_color.py -m gpu -q -x > gpurun_out/r2c7_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r2c7_pytest.log
tail -3 gpurun_out/r2c7_pytest.log
L=variants/libfsr1_
timeout 500 python tools/abtest.py --libs ${L}oldring.so,${L}newring.so,${L}nr_s3.so,${L}nr_s4d4.so,${L}nr_d4.so --workloads 1080p_to_4k --kernels rcas,pair --reps 3 > gpurun_out/r2c7_ab.log 2>&1
cat gpurun_out/r2c7_ab.log
timeout 400 python tools/abtest.py --libs ${L}oldring.so,${L}newring.so,${L}nr_s3.so,${L}nr_s4d4.so,${L}nr_d4.so --workloads 1440p_to_4k_x8,4k_to_8k_x16,540p_to_1080p --kernels rcas,pair --reps 1 --launches 200 > gpurun_out/r2c7_ab2.log 2>&1
cat gpurun_out/r2c7_ab2.log
