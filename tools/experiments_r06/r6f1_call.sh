timeout 900 python -m pytest tests/test_gpu_strict.py tests/test_gpu_image_parity.py -m gpu -x -q 2>&1 | tail -5
timeout 900 python tools/abtest.py --libs variants/libfsr1_base.so@0x40,@0x40 --workloads 1080p_to_4k,1440p_to_4k,1662p_to_4k,540p_to_1080p,4k_to_8k --kernels easu,pair --reps 3 2>&1 | tee gpurun_out/r6f1_strict_fixup.log
