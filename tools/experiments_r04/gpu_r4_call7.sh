#!/bin/bash
# call 7: pipeline tests (colour stages / UNORM / batches / graph capture), launch-bound sizes from the C host and with hipGraphs.
cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_pipeline.py -x -q -m gpu 2>&1 | tail -5 | tee gpurun_out/r4c7_pytest.log
R=runner/fsr1_runner; O=gpurun_out/r4c7_small_frames.log; : > $O
for S in 1 2 3; do
timeout 120 $R --gpus 1 --in 960x540 --out 1920x1080 --steps 5000 --warmup 200 --streams $S >> $O 2>&1
timeout 120 $R --gpus 1 --in 960x540 --out 1920x1080 --steps 5000 --warmup 200 --pipeline auto --streams $S >> $O 2>&1
timeout 120 $R --gpus 1 --in 1280x720 --out 1920x1080 --steps 5000 --warmup 200 --pipeline auto --streams $S >> $O 2>&1
timeout 120 $R --gpus 1 --in 1280x720 --out 2560x1440 --steps 5000 --warmup 200 --pipeline auto --streams $S >> $O 2>&1
done
grep '^{' $O | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('runner', d['in'], d['out'], d['pipeline'], d['pipeline_run'], 'streams', d['streams'], '->', d['value'], 'Mpix/s', round(d['ms_per_step'] * 1e3, 2), 'us/frame')
" | tee -a gpurun_out/r4c7_small_frames.log
for G in 0 8; do for S in 1 2; do
timeout 200 python bench.py --workload 540p_to_1080p --pipeline fused --streams $S --graph $G --no-cpu-baseline --no-also --steps 4000 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('bench 540p_to_1080p fused streams', d['config']['streams'], 'graph', d['config']['hip_graph_steps'], '->', d['value'], 'Mpix/s', round(d['ms_per_step'] * 1e3, 2), 'us/frame')
" | tee -a gpurun_out/r4c7_small_frames.log
done; done
