#!/bin/bash
# Round 5, GPU call 3: row-per-wave staging in every EASU-class kernel (tree) vs the round-4 staging (variants/libfsr1_base.so); full suite.
O=gpurun_out/r5c3; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q -x --durations=5 > $O/pytest_gpu.log 2>&1; echo "rc=$?" >> $O/pytest_gpu.log; tail -4 $O/pytest_gpu.log
python tools/abtest.py --libs variants/libfsr1_base.so,fidelityfx-fsr_amd/libfsr1_hip.so --workloads 1080p_to_4k,1440p_to_4k,1662p_to_4k,831p_to_1080p,540p_to_1080p,4k_to_8k --kernels easu,pair,fused --reps 3 > $O/ab_tree.log 2>&1; cat $O/ab_tree.log
python tools/abtest.py --libs variants/libfsr1_base.so,fidelityfx-fsr_amd/libfsr1_hip.so --workloads 1080p_to_4k,1440p_to_4k --kernels easu,fused --math exact --reps 2 > $O/ab_tree_exact.log 2>&1; cat $O/ab_tree_exact.log
python tools/abtest.py --libs variants/libfsr1_base.so,fidelityfx-fsr_amd/libfsr1_hip.so --workloads 4k_to_8k_x16,1440p_to_4k_x8 --kernels easu,fused --reps 2 --launches 60 > $O/ab_tree_batch.log 2>&1; cat $O/ab_tree_batch.log
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_k20.json 2> $O/bench_k20.err; cut -c1-300 $O/bench_k20.json
