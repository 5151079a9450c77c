#!/bin/bash
# Round 5, GPU call 7: the generic EASU kernel on 64 x 32 tiles with 512-thread workgroups (tree) vs 64 x 16 / 256 threads (variants/libfsr1_stage.so)
O=gpurun_out/r5c7; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_image_parity.py tests/test_gpu_fullframe.py -q -x -k "generic_tall or true_ratio or reference_chain or 1440p or dynamic_resolution or tall" > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; tail -3 $O/pytest.log
python tools/abtest.py --libs variants/libfsr1_stage.so,fidelityfx-fsr_amd/libfsr1_hip.so --workloads 1440p_to_4k,1270p_to_4k,1662p_to_4k,831p_to_1080p,720p_to_1080p,1440p_to_4k_x8 --kernels easu,pair --reps 3 > $O/ab_generic_tall.log 2>&1; cat $O/ab_generic_tall.log
# small outputs both ways (the rule's threshold): the test library with the tall tiles forced on / off
python tools/abtest.py --libs "fidelityfx-fsr_amd/libfsr1_hip.so%FSR1_EASU_TALL=0,fidelityfx-fsr_amd/libfsr1_hip.so%FSR1_EASU_TALL=1" --workloads 831p_to_1080p,720p_to_1080p,1440p_to_4k --kernels easu --reps 2 > $O/ab_generic_tall_small.log 2>&1; cat $O/ab_generic_tall_small.log
