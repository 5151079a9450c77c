#!/bin/bash
# call 1 (round 4): the whole -m gpu suite on the tree (new: source-level Hx2 / ARmp8x8 shell, fused exact-2x H kernel, RCAS-H
# over-fetch fix, band intermediary check, run-steps hook), the bench line at the driver's K = 20 and at the default K (stopwatch
# self-consistency), A/B of the H kernels against round 3's library, and the fused launch's per-workgroup timeline.
cd /root/repo
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r4c1_pytest.log; tail -3 gpurun_out/r4c1_pytest.log
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r4c1_bench_k20.json 2> gpurun_out/r4c1_bench_k20.err; cut -c1-600 gpurun_out/r4c1_bench_k20.json
timeout 300 python bench.py > gpurun_out/r4c1_bench_default.json 2> gpurun_out/r4c1_bench_default.err; cut -c1-300 gpurun_out/r4c1_bench_default.json
timeout 600 python tools/abtest.py --libs "variants/libfsr1_r3.so,%FSR1_AB_DEFAULT=1" --workloads 1080p_to_4k --kernels easu,rcas,rcas_cold,pair,fused --math h --reps 3 2>&1 | tee gpurun_out/r4c1_h_ab.log
timeout 600 python tools/abtest.py --libs "variants/libfsr1_r3.so,%FSR1_AB_DEFAULT=1" --workloads 4k_to_8k_x16,1080p_to_4k_x4 --kernels pair,fused --math h --reps 2 --launches 200 2>&1 | tee -a gpurun_out/r4c1_h_ab.log
timeout 600 python tools/abtest.py --libs "variants/libfsr1_r3.so,%FSR1_AB_DEFAULT=1" --workloads 1080p_to_4k,1440p_to_4k --kernels easu,rcas,pair,fused --reps 2 2>&1 | tee gpurun_out/r4c1_f_ab.log
timeout 300 python tools/experiments_r04/fused_trace.py > gpurun_out/r4c1_fused_trace.log 2>&1; tail -40 gpurun_out/r4c1_fused_trace.log
