#!/bin/bash
# call 23: walking fused kernel with the next step's texels fetched before the RCAS phase, at six waves per SIMD (77 VGPRs,
# no spills) against the tree (seven waves, no prefetch) — parity first
cd /root/repo
mkdir -p gpurun_out
FSR1_HIP_LIB=$PWD/variants/libfsr1_pf6.so FSR1_FUSED_S2_STEPS=3 timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fused" --deselect tests/test_gpu_parity.py::test_fused_exact_2x_run_steps 2>&1 | tail -2 | tee gpurun_out/r3c23_pytest.log
timeout 900 python tools/abtest.py --libs "%FSR1_AB_DEFAULT=1,variants/libfsr1_pf6.so" --workloads 4k_to_8k_x16,1080p_to_4k_x4 --kernels fused --reps 3 2>&1 | tee gpurun_out/r3c23_prefetch_six_waves.log
timeout 300 python tools/abtest.py --libs "%FSR1_FUSED_S2_STEPS=2,variants/libfsr1_pf6.so%FSR1_FUSED_S2_STEPS=2" --workloads 1080p_to_4k --kernels fused --reps 2 2>&1 | tee -a gpurun_out/r3c23_prefetch_six_waves.log
