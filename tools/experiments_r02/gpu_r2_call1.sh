#!/bin/bash
# round-2 GPU call 1: full gpu test suite (incl. whole-frame parity vs oracle/_ref) + A/B of EASU/RCAS variants
mkdir -p gpurun_out
ls -la oracle/_ref/ > gpurun_out/r2c1_env.log 2>&1
nproc >> gpurun_out/r2c1_env.log
timeout 900 python -m pytest tests -m gpu -q --maxfail=20 -x --durations=15 > gpurun_out/r2c1_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r2c1_pytest.log
tail -5 gpurun_out/r2c1_pytest.log
timeout 400 python tools/abtest.py --libs variants/libfsr1_r01.so,variants/libfsr1_cur.so,variants/libfsr1_oneway.so,variants/libfsr1_ring4.so,variants/libfsr1_rows32r4.so,variants/libfsr1_rows8.so,variants/libfsr1_ring4w1.so --workloads 1080p_to_4k --reps 2 > gpurun_out/r2c1_ab_4k.log 2>&1
timeout 300 python tools/abtest.py --libs variants/libfsr1_r01.so,variants/libfsr1_cur.so --workloads 1440p_to_4k,1662p_to_4k,1440p_to_4k_x8,4k_to_8k_x16 --reps 2 --launches 200 > gpurun_out/r2c1_ab_ratio.log 2>&1
cat gpurun_out/r2c1_ab_4k.log gpurun_out/r2c1_ab_ratio.log
