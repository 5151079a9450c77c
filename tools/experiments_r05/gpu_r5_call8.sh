#!/bin/bash
# Round 5, GPU call 8: batches as ONE launch (pair) per step vs the same frames as single-frame submissions through the pipeline
O=gpurun_out/r5c8; mkdir -p $O
for w in 4k_to_8k 4k_to_8k_x16; do for p in fused two-pass; do
python bench.py --workload $w --pipeline $p --no-cpu-baseline --no-also --no-latency --no-parity --steps $([ $w = 4k_to_8k ] && echo 600 || echo 40) --warmup 10 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); print('$w $p', 'pipelined', d['value'], d['ms_per_step'], 'one_stream', d['one_stream'], 'steady', d.get('steady_state',{}).get('value'))" >> $O/batch_vs_frames.log
done; done
for w in 1440p_to_4k 1440p_to_4k_x8; do
python bench.py --workload $w --no-cpu-baseline --no-also --no-latency --no-parity --steps $([ $w = 1440p_to_4k ] && echo 1600 || echo 200) --warmup 10 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); print('$w two-pass', 'pipelined', d['value'], d['ms_per_step'], 'one_stream', d['one_stream'], 'steady', d.get('steady_state',{}).get('value'))" >> $O/batch_vs_frames.log
done
cat $O/batch_vs_frames.log
