set -x
timeout 900 python -m pytest tests -m gpu -x -q -k "rcas or fullframe or bands or unorm or image_parity or pipeline" 2>&1 | tail -5
timeout 600 python tools/abtest.py --libs variants/libfsr1_base.so,@0x0 --workloads 1080p_to_4k,1440p_to_4k_x8,4k_to_8k --kernels rcas,rcas_cold,pair --reps 3 2>&1 | tee gpurun_out/r6d1_rcas_updown.log
