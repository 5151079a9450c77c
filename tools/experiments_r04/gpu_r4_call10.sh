#!/bin/bash
# call 10: the 64 x 32 EASU tile (F kernels) alone and under pipelining; parity of the variant first.
cd /root/repo
mkdir -p gpurun_out
FSR1_HIP_LIB=$PWD/variants/libfsr1_tile32.so timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullframe.py -x -q -m gpu -k "not packed and not _h_ and not fp16" 2>&1 | tail -3 | tee gpurun_out/r4c10_tile32.log
timeout 300 python tools/abtest.py --libs "variants/libfsr1_tile32.so,%FSR1_AB_DEFAULT=1" --workloads 1080p_to_4k,1440p_to_4k,4k_to_8k_x16 --kernels easu,pair --reps 3 --launches 300 2>&1 | tee -a gpurun_out/r4c10_tile32.log
FSR1_HIP_LIB=$PWD/variants/libfsr1_tile32.so timeout 300 python tools/experiments_r04/streams_sweep.py 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r4c10_tile32.log
