#!/bin/bash
# call 2: the tall (512-thread, 62 x 30) one-step tile of the exact-2x fused launch — parity of the new tests, then A/B against the
# 256-thread tile at several sizes; independent frames on alternating streams.
cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity_h.py tests/test_gpu_fullframe.py tests/test_gpu_bands.py -x -q -m gpu -k "fused or tall or band" 2>&1 | tail -4 | tee gpurun_out/r4c2_pytest.log
timeout 600 python tools/abtest.py --libs "%FSR1_FUSED_S2_TALL=0,%FSR1_FUSED_S2_TALL=1,%FSR1_AB_DEFAULT=1" --workloads 1080p_to_4k,720p_to_1440p,540p_to_1080p --kernels fused --reps 3 2>&1 | tee gpurun_out/r4c2_tall_ab.log
timeout 600 python tools/abtest.py --libs "%FSR1_AB_DEFAULT=1,%FSR1_FUSED_S2_STEPS=1" --workloads 1080p_to_4k_x2,1080p_to_4k_x4,4k_to_8k --kernels fused --reps 3 --launches 200 2>&1 | tee -a gpurun_out/r4c2_tall_ab.log
timeout 600 python tools/abtest.py --libs "@0x10%FSR1_FUSED_S2_TALL=0,@0x10%FSR1_FUSED_S2_TALL=1" --workloads 1080p_to_4k --kernels fused --reps 2 2>&1 | tee -a gpurun_out/r4c2_tall_ab.log
timeout 600 python tools/experiments_r04/two_stream.py 2>&1 | tee gpurun_out/r4c2_two_stream.log
