#!/bin/bash
# call 3: the frame pipeline (fsr1_pipeline: steps on alternating streams) — parity tests, the bench line at K = 20 and default with
# --streams 2 (default) and 1, the C runner with --streams 1 / 2.
cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_pipeline.py tests/test_shard.py tests/test_runner.py tests/test_abi.py -x -q -m gpu 2>&1 | tail -4 | tee gpurun_out/r4c3_pytest.log
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r4c3_bench_k20.json 2> gpurun_out/r4c3_bench_k20.err; cut -c1-250 gpurun_out/r4c3_bench_k20.json; tail -3 gpurun_out/r4c3_bench_k20.err
timeout 300 python bench.py > gpurun_out/r4c3_bench_default.json 2> gpurun_out/r4c3_bench_default.err; cut -c1-250 gpurun_out/r4c3_bench_default.json
timeout 300 python bench.py --streams 1 --no-cpu-baseline > gpurun_out/r4c3_bench_one_stream.json 2> gpurun_out/r4c3_bench_one_stream.err; cut -c1-250 gpurun_out/r4c3_bench_one_stream.json
R=runner/fsr1_runner; O=gpurun_out/r4c3_runner.log; : > $O
for S in 1 2; do
timeout 300 $R --gpus 1 --steps 2000 --warmup 100 --streams $S >> $O 2>&1
timeout 300 $R --gpus 1 --steps 2000 --warmup 100 --pipeline auto --streams $S >> $O 2>&1
timeout 300 $R --gpus 1 --in 2560x1440 --out 3840x2160 --steps 1000 --warmup 100 --streams $S >> $O 2>&1
timeout 300 $R --gpus 1 --frames 8 --in 2560x1440 --out 3840x2160 --steps 100 --warmup 10 --streams $S >> $O 2>&1
timeout 300 $R --gpus 1 --frames 16 --in 3840x2160 --out 7680x4320 --steps 30 --warmup 5 --pipeline auto --streams $S >> $O 2>&1
timeout 300 $R --gpus 1 --steps 2000 --warmup 100 --math h --streams $S >> $O 2>&1
done
grep '^{' $O | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['in'], d['out'], 'frames/step', d['frames'] // d['steps'], d['pipeline'], d['pipeline_run'], d['math'], 'streams', d['streams'], '->', d['value'], 'Mpix/s', d['ms_per_step'], 'ms/step')
"
