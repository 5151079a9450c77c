#!/bin/bash
# call 19: one-step form compiled separately; steps sweep at the sizes between one 4K frame and the 16-frame 8K batch
cd /root/repo
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fused" 2>&1 | tail -3 > gpurun_out/r3c19_pytest.log
cat gpurun_out/r3c19_pytest.log
L=variants/libfsr1_base.so
timeout 900 python tools/abtest.py --libs "$L,%FSR1_FUSED_S2_STEPS=1" --workloads 1080p_to_4k,540p_to_1080p --kernels fused --reps 3 2>&1 | tee gpurun_out/r3c19_one_step.log
timeout 900 python tools/abtest.py --libs "$L,%FSR1_FUSED_S2_STEPS=2,%FSR1_FUSED_S2_STEPS=3,%FSR1_FUSED_S2_STEPS=4,%FSR1_FUSED_S2_STEPS=6,%FSR1_FUSED_S2_STEPS=8" \
   --workloads 4k_to_8k,1080p_to_4k_x4 --kernels fused --reps 2 2>&1 | tee gpurun_out/r3c19_steps_mid.log
