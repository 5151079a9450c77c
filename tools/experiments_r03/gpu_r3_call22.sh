#!/bin/bash
# call 22: start-up stagger of the first residency's workgroups in the walking fused kernel (single 4K frame)
cd /root/repo
mkdir -p gpurun_out
L=variants/libfsr1_base.so
V=""
for s in 2 3 4 5; do for g in 0 1 2; do V="$V,%FSR1_FUSED_S2_STEPS=$s%FSR1_FUSED_S2_STAGGER=$g"; done; done
timeout 900 python tools/abtest.py --libs "$L$V" --workloads 1080p_to_4k --kernels fused --reps 2 2>&1 | tee gpurun_out/r3c22_stagger_4k.log
