#!/bin/bash
# Round 5, GPU call 6: how sensitive the generic EASU kernel is to workgroups per CU (dynamic LDS padded by 6 / 12 KB): is shrinking the
# LDS record (32 -> 24 B per texel) worth building for the ratios near 1x, which are LDS-bound at 4-5 workgroups per CU?
O=gpurun_out/r5c6; mkdir -p $O
python tools/abtest.py --libs fidelityfx-fsr_amd/libfsr1_hip.so,variants/libfsr1_ldspad6144.so,variants/libfsr1_ldspad12288.so --workloads 1440p_to_4k,1662p_to_4k,1270p_to_4k,1080p_to_4k --kernels easu --reps 2 > $O/ab_ldspad.log 2>&1; cat $O/ab_ldspad.log
