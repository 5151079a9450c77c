#!/bin/bash
# round-2 GPU call 9: RCAS-H strip shape (F-kernel shape, rolled row loop) — parity + A/B
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity_h.py -m gpu -q -x > gpurun_out/r2c9_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r2c9_pytest.log
tail -3 gpurun_out/r2c9_pytest.log
L=variants/libfsr1_
timeout 500 python tools/abtest.py --libs ${L}hold.so,${L}h16w2.so,${L}h24w4.so,${L}h16w4.so,${L}h8w2.so,${L}h32w2.so --workloads 1080p_to_4k,540p_to_1080p --math h --kernels rcas,pair --reps 2 > gpurun_out/r2c9_ab.log 2>&1
cat gpurun_out/r2c9_ab.log
