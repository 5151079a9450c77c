#!/bin/bash
# Round 5, GPU call 10: exact-2x fused batches as one walking launch vs frame by frame through the pipeline
O=gpurun_out/r5c10; mkdir -p $O
for w in 1080p_to_4k 1080p_to_4k_x4 1080p_to_4k_x8 540p_to_1080p; do
python bench.py --workload $w --pipeline fused --no-cpu-baseline --no-also --no-latency --no-parity --steps $([ $w = 1080p_to_4k_x8 ] && echo 200 || ([ $w = 1080p_to_4k_x4 ] && echo 400 || echo 1600)) --warmup 10 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); print('$w fused', 'pipelined', d['value'], d['ms_per_step'], 'one_stream', d['one_stream'], 'steady', d.get('steady_state',{}).get('value'))" >> $O/fused_batch_vs_frames.log
done
cat $O/fused_batch_vs_frames.log
