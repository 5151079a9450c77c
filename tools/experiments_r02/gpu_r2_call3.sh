#!/bin/bash
# round-2 GPU call 3: store-policy tests, RCAS geometry sweep with streaming stores, fused/pair re-measure
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity_h.py tests/test_abi.py tests/test_gpu_unorm.py tests/test_gpu_color.py -m gpu -q -x > gpurun_out/r2c3_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r2c3_pytest.log
tail -4 gpurun_out/r2c3_pytest.log
L=variants/libfsr1_
timeout 500 python tools/abtest.py --libs fidelityfx-fsr_amd/libfsr1_hip.so,${L}rows8.so,${L}rows32.so,${L}w4.so,${L}w1.so,${L}r4.so --workloads 1080p_to_4k --kernels easu,rcas,pair,fused --reps 2 > gpurun_out/r2c3_ab.log 2>&1
cat gpurun_out/r2c3_ab.log
timeout 300 python tools/abtest.py --libs fidelityfx-fsr_amd/libfsr1_hip.so,${L}shallow.so,${L}rows8.so --workloads 1440p_to_4k,1440p_to_4k_x8,4k_to_8k_x16,540p_to_1080p --kernels easu,rcas,pair,fused --reps 1 --launches 200 > gpurun_out/r2c3_ab2.log 2>&1
cat gpurun_out/r2c3_ab2.log
