#!/bin/bash
# round-2 GPU call 6: RCAS strip height on batches (apron re-reads vs wave count); 1.3x pipelines after the store policy
mkdir -p gpurun_out
L=variants/libfsr1_
timeout 600 python tools/abtest.py --libs fidelityfx-fsr_amd/libfsr1_hip.so,${L}rows24.so,${L}rows32.so,${L}rows64.so --workloads 1440p_to_4k_x8,4k_to_8k_x16 --kernels rcas,pair --reps 2 --launches 200 > gpurun_out/r2c6_ab.log 2>&1
cat gpurun_out/r2c6_ab.log
timeout 200 python tools/abtest.py --libs fidelityfx-fsr_amd/libfsr1_hip.so --workloads 1662p_to_4k,1270p_to_4k,270p_to_540p --kernels easu,rcas,pair,fused --reps 1 > gpurun_out/r2c6_ab2.log 2>&1
cat gpurun_out/r2c6_ab2.log
