#!/bin/bash
# call 17: the column-walking fused exact-2x kernel — parity, then steps-per-run sweep against the committed kernel
cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fused" 2>&1 | tail -5 > gpurun_out/r3c17_pytest.log
timeout 600 python -m pytest tests/test_gpu_bands.py tests/test_shard.py -x -q -m gpu 2>&1 | tail -5 >> gpurun_out/r3c17_pytest.log
cat gpurun_out/r3c17_pytest.log
L=variants/libfsr1_base.so
timeout 900 python tools/abtest.py --libs "$L,%FSR1_FUSED_S2_STEPS=1,%FSR1_FUSED_S2_STEPS=2,%FSR1_FUSED_S2_STEPS=3,%FSR1_FUSED_S2_STEPS=4,%FSR1_FUSED_S2_STEPS=5,%FSR1_FUSED_S2_STEPS=6,%FSR1_FUSED_S2_STEPS=8,%FSR1_AB_DEFAULT=1" \
   --workloads 1080p_to_4k --kernels fused --reps 2 2>&1 | tee gpurun_out/r3c17_steps_4k.log
timeout 900 python tools/abtest.py --libs "$L,%FSR1_FUSED_S2_STEPS=4,%FSR1_FUSED_S2_STEPS=8,%FSR1_FUSED_S2_STEPS=16,%FSR1_AB_DEFAULT=1" \
   --workloads 4k_to_8k_x16,540p_to_1080p --kernels fused --reps 2 2>&1 | tee gpurun_out/r3c17_steps_other.log
