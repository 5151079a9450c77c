#!/bin/bash
# round-2 GPU call 4: RCAS VALU cuts (asm min/max, scalar row bases, 3x unroll) — parity + A/B
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_unorm.py tests/test_gpu_color.py tests/test_device_api.py -m gpu -q -x > gpurun_out/r2c4_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r2c4_pytest.log
tail -4 gpurun_out/r2c4_pytest.log
L=variants/libfsr1_
timeout 400 python tools/abtest.py --libs ${L}prevrcas.so,${L}cur.so,${L}u3.so,${L}u3r12.so --workloads 1080p_to_4k --kernels rcas,pair,fused --reps 3 > gpurun_out/r2c4_ab.log 2>&1
cat gpurun_out/r2c4_ab.log
timeout 300 python tools/abtest.py --libs ${L}prevrcas.so,${L}cur.so,${L}u3.so --workloads 1440p_to_4k_x8,4k_to_8k_x16 --kernels rcas,pair --reps 1 --launches 200 > gpurun_out/r2c4_ab2.log 2>&1
cat gpurun_out/r2c4_ab2.log
