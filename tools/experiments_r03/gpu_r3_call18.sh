#!/bin/bash
# call 18: column-walking fused kernel with phase 4 unrolled by hand — steps sweep again
cd /root/repo
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fused" 2>&1 | tail -3 > gpurun_out/r3c18_pytest.log
cat gpurun_out/r3c18_pytest.log
L=variants/libfsr1_base.so
timeout 900 python tools/abtest.py --libs "$L,%FSR1_FUSED_S2_STEPS=1,%FSR1_FUSED_S2_STEPS=2,%FSR1_FUSED_S2_STEPS=3,%FSR1_FUSED_S2_STEPS=5" \
   --workloads 1080p_to_4k --kernels fused --reps 3 2>&1 | tee gpurun_out/r3c18_steps_4k.log
timeout 900 python tools/abtest.py --libs "$L,%FSR1_FUSED_S2_STEPS=1,%FSR1_FUSED_S2_STEPS=2,%FSR1_FUSED_S2_STEPS=6,%FSR1_FUSED_S2_STEPS=8,%FSR1_FUSED_S2_STEPS=10" \
   --workloads 4k_to_8k_x16 --kernels fused --reps 2 2>&1 | tee gpurun_out/r3c18_steps_8k.log
