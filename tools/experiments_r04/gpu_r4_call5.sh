#!/bin/bash
# call 5: FSR1_FLAG_FRAMES_OVERLAP (the pipeline's fused launches walk) — parity, then the H pipelines and the RCAS strip height under
# two-stream pipelining, then the bench line.
cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_pipeline.py tests/test_runner.py -x -q -m gpu 2>&1 | tail -3 | tee gpurun_out/r4c5_pytest.log
timeout 300 python tools/experiments_r04/two_stream_h.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4c5_two_stream_h.log
timeout 300 python tools/experiments_r04/two_stream_h.py f 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4c5_two_stream_rcas_rows.log
FSR1_HIP_LIB=$PWD/variants/libfsr1_rcas16.so timeout 300 python tools/experiments_r04/two_stream_h.py f 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r4c5_two_stream_rcas_rows.log
timeout 300 python bench.py --no-cpu-baseline > gpurun_out/r4c5_bench_default.json 2> gpurun_out/r4c5_bench_default.err; python - <<'PY'
import json
d = json.load(open('gpurun_out/r4c5_bench_default.json'))
print(d['value'], d['ms_per_step'], d['config']['one_stream'], d['stopwatch_suspect'])
print({k: (v['value'], v['ms_per_step'], v.get('one_stream', {}).get('value')) for k, v in d['also_measured'].items()})
PY
