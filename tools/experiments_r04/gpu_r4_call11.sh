#!/bin/bash
# call 11: 64 x 32 EASU tiles in the tree (exact 2x, default arithmetic; rule: large launches or FSR1_FLAG_FRAMES_OVERLAP) — parity,
# the launch size from which they pay alone, and the pipelines of 1 .. 4 streams with the rule in place.
cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullframe.py tests/test_gpu_pipeline.py tests/test_gpu_bands.py tests/test_gpu_unorm.py -x -q -m gpu 2>&1 | tail -3 | tee gpurun_out/r4c11_pytest.log
timeout 600 python tools/abtest.py --libs "%FSR1_EASU_TALL=0,%FSR1_EASU_TALL=1" --workloads 1080p_to_4k,1080p_to_4k_x2,1080p_to_4k_x4,4k_to_8k,1080p_to_4k_x8,4k_to_8k_x16,720p_to_1440p --kernels easu,pair --reps 2 --launches 300 2>&1 | tee gpurun_out/r4c11_easu_tall_ab.log
timeout 600 python tools/experiments_r04/streams_sweep.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4c11_streams_sweep.log
