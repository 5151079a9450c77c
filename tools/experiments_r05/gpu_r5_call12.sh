#!/bin/bash
# Round 5, GPU call 12: the compact pitched LDS layout of the generic EASU kernel (24 B per footprint texel: (dirX, dirY) in 8-byte analysis
# records, lenX^2 + lenY^2 in the texel's own .w) vs 32 B per texel (variants/libfsr1_head.so)
O=gpurun_out/r5c12; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_image_parity.py tests/test_gpu_fullframe.py tests/test_gpu_bands.py tests/test_gpu_unorm.py -q -x -k "generic or true_ratio or reference_chain or 1440p or dynamic_resolution or golden or oracle or random or band or rgba8 or rgb10" > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; tail -3 $O/pytest.log
python tools/abtest.py --libs variants/libfsr1_head.so,fidelityfx-fsr_amd/libfsr1_hip.so --workloads 1440p_to_4k,1270p_to_4k,1662p_to_4k,831p_to_1080p,720p_to_1080p,1440p_to_4k_x8 --kernels easu,pair --reps 3 > $O/ab_ana8.log 2>&1; cat $O/ab_ana8.log
