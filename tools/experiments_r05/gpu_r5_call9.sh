#!/bin/bash
# Round 5, GPU call 9: batches split into frames by the pipeline — tests, the bench's configs[2] shard, the C runner
O=gpurun_out/r5c9; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_pipeline.py tests/test_runner.py tests/test_gpu_fullframe.py -m gpu -q -x -k "pipeline or batch or runner" > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; tail -3 $O/pytest.log
for w in 1440p_to_4k_x8 1440p_to_4k 1080p_to_4k_x8; do
python bench.py --workload $w --no-cpu-baseline --no-also --no-latency --no-parity --steps $([ $w = 1440p_to_4k ] && echo 1600 || echo 200) --warmup 10 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); print('$w two-pass', 'pipelined', d['value'], d['ms_per_step'], 'one_stream', d['one_stream'], 'steady', d.get('steady_state',{}).get('value'))" >> $O/batch_split.log
done
R=runner/fsr1_runner
for S in 3 1; do timeout 300 $R --gpus 1 --frames 8 --in 2560x1440 --out 3840x2160 --steps 100 --warmup 10 --streams $S 2>/dev/null | cut -c1-160 >> $O/batch_split.log; done
cat $O/batch_split.log
