timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "seam" 2>&1 | tail -5
timeout 900 python tools/abtest.py --libs @0x40,@0x40%FSR1_FUSED_S2_TALL=1,@0x40%FSR1_FUSED_SX=1,@0x40%FSR1_FUSED_SX=1%FSR1_FUSED_S2_TALL=0 --workloads 1080p_to_4k,540p_to_1080p,4k_to_8k --kernels pair,fused --reps 2 2>&1 | tee gpurun_out/r6e3_fused_sx_strict.log
