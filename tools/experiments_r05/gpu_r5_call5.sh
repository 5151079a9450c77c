#!/bin/bash
# Round 5, GPU call 5: where the packed-fp16 exact-2x fused launch overtakes the two H dispatches on batches
O=gpurun_out/r5c5; mkdir -p $O
python tools/abtest.py --libs fidelityfx-fsr_amd/libfsr1_hip.so --workloads 1080p_to_4k_x8,4k_to_8k_x4,4k_to_8k_x8,4k_to_8k_x16,720p_to_1440p,1080p_to_4k --kernels pair,fused --math h --reps 3 --launches 60 > $O/ab_h_auto2.log 2>&1; cat $O/ab_h_auto2.log
