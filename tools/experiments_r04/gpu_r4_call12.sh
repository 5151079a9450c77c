#!/bin/bash
# call 12: under three-stream pipelining — RCAS strips of 32 rows; the generic (1.5x) EASU kernel on 64 x 32 tiles.
cd /root/repo
mkdir -p gpurun_out
export SWEEP_QUICK=1
for rep in 1 2; do
timeout 200 python tools/experiments_r04/streams_sweep.py 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r4c12_overlap_geometry.log
FSR1_HIP_LIB=$PWD/variants/libfsr1_rcas32.so timeout 200 python tools/experiments_r04/streams_sweep.py 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r4c12_overlap_geometry.log
FSR1_EASU_TALL=0 FSR1_HIP_LIB=$PWD/variants/libfsr1_tile32.so timeout 200 python tools/experiments_r04/streams_sweep.py 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r4c12_overlap_geometry.log
done
