timeout 300 python -m pytest tests/test_gpu_strict.py -m gpu -x -q 2>&1 | tail -3
timeout 900 python tools/abtest.py --libs variants/libfsr1_base.so@0x40,@0x40 --workloads 1080p_to_4k,1440p_to_4k,540p_to_1080p --kernels easu,pair --reps 2 2>&1 | tee gpurun_out/r6f2_strict_fixup.log
for lib in variants/libfsr1_base.so ""; do
  FSR1_HIP_LIB=${lib:+$PWD/$lib} timeout 300 python bench.py --steps 2000 --warmup 200 --no-cpu-baseline --no-also --no-latency --no-parity --no-submit-ceiling 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib', d['value'], d['ms_per_step'], d.get('one_stream'), d.get('steady_state'))" | tee -a gpurun_out/r6f2_strict_fixup.log
done
