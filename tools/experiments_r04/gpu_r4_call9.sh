#!/bin/bash
# call 9: how many streams (1 .. 4) per workload; the 64 x 32 EASU tile alone and under pipelining.
cd /root/repo
mkdir -p gpurun_out
timeout 600 python tools/experiments_r04/streams_sweep.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4c9_streams_sweep.log
FSR1_HIP_LIB=$PWD/variants/libfsr1_tile32.so timeout 300 python tools/experiments_r04/streams_sweep.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4c9_tile32.log
timeout 300 python tools/abtest.py --libs "variants/libfsr1_tile32.so,%FSR1_AB_DEFAULT=1" --workloads 1080p_to_4k,4k_to_8k_x16 --kernels easu,pair --reps 3 --launches 300 2>&1 | tee -a gpurun_out/r4c9_tile32.log
