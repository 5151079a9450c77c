#!/bin/bash
# round-2 GPU call 2: test suite on the refactored tree (device API, H exact-2x), RCAS variants A/B, H timings
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -x --durations=8 > gpurun_out/r2c2_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r2c2_pytest.log
tail -14 gpurun_out/r2c2_pytest.log
L=variants/libfsr1_
timeout 500 python tools/abtest.py --libs ${L}e41.so,${L}e41alt.so,${L}e41r4.so,${L}e41altr4.so,${L}e41nt.so,${L}e41ntl.so,${L}e41altnt.so --workloads 1080p_to_4k --kernels rcas,pair --reps 3 > gpurun_out/r2c2_ab_rcas.log 2>&1
cat gpurun_out/r2c2_ab_rcas.log
timeout 200 python tools/abtest.py --libs ${L}e41.so,${L}e41altnt.so --workloads 1440p_to_4k_x8,4k_to_8k_x16 --kernels rcas,pair --reps 1 --launches 200 > gpurun_out/r2c2_ab_rcas_batch.log 2>&1
cat gpurun_out/r2c2_ab_rcas_batch.log
timeout 200 python tools/abtest.py --libs variants/libfsr1_r01.so,fidelityfx-fsr_amd/libfsr1_hip.so --workloads 1080p_to_4k,1440p_to_4k --math h --kernels easu,rcas,pair --reps 2 > gpurun_out/r2c2_ab_h.log 2>&1
cat gpurun_out/r2c2_ab_h.log
