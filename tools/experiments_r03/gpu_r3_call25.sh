#!/bin/bash
# call 25: RCAS edge wave-columns on the interior body with the one outside apron texel replaced (XEDGE) — parity, then A/B
cd /root/repo
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bands.py tests/test_gpu_unorm.py tests/test_gpu_special_values.py tests/test_gpu_color.py -x -q -m gpu -k "rcas or band or unorm or special or color or two_pass or upscale" 2>&1 | tail -3 | tee gpurun_out/r3c25_pytest.log
timeout 600 python tools/abtest.py --libs "variants/libfsr1_base.so,%FSR1_AB_DEFAULT=1" --workloads 1080p_to_4k,1440p_to_4k_x8,540p_to_1080p --kernels rcas,rcas_cold,pair --reps 3 2>&1 | tee gpurun_out/r3c25_rcas_xedge_ab.log
