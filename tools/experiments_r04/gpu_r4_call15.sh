#!/bin/bash
# call 15: the generic EASU kernel keeps analyses and dering bounds across a lane's consecutive rows that share their 'f' row
cd /root/repo
mkdir -p gpurun_out
FSR1_HIP_LIB=$PWD/variants/libfsr1_colreuse.so timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullframe.py tests/test_gpu_extents.py tests/test_gpu_bands.py tests/test_gpu_special_values.py -x -q -m gpu 2>&1 | tail -3 | tee gpurun_out/r4c15_colreuse.log
timeout 600 python tools/abtest.py --libs "variants/libfsr1_colreuse.so,%FSR1_AB_DEFAULT=1" --workloads 1440p_to_4k,1270p_to_4k,1662p_to_4k,1440p_to_4k_x8,720p_to_1080p --kernels easu,pair --reps 3 --launches 300 2>&1 | tee -a gpurun_out/r4c15_colreuse.log
timeout 300 python tools/abtest.py --libs "variants/libfsr1_colreuse.so@0x10,@0x10" --workloads 1440p_to_4k --kernels easu --reps 2 --launches 300 2>&1 | tee -a gpurun_out/r4c15_colreuse.log
