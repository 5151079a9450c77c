#!/bin/bash
# Round 5, GPU call 11: every bench.py mode once (the new blocks must not break the optional modes)
O=gpurun_out/r5c11; mkdir -p $O; : > $O/modes.log
run() { echo "== $*" >> $O/modes.log; timeout 300 python bench.py --steps 40 --warmup 5 "$@" 2>$O/err.txt | grep '^{' | python -c "
import sys,json
l=sys.stdin.readline()
if not l: print('NO LINE'); sys.exit()
d=json.loads(l); print(d['value'], d['ms_per_step'], d['parity_class'], 'one', d['one_stream']['value'], 'steady' in str(d.get('steady_state')), 'lat' , 'latency_us' in d, 'parity', (d.get('parity') or {}).get('two_dispatch',{}).get('frac_within_1ulp'), 'suspect', d.get('stopwatch_suspect'))" >> $O/modes.log; tail -2 $O/err.txt >> $O/modes.log; }
run --graph 6 --workload 270p_to_540p --no-cpu-baseline --no-also
run --pipeline easu --workload 540p_to_1080p --storage rgba32f --no-also
run --pipeline color --stages 7 --no-cpu-baseline
run --pipeline fused --stages 7 --no-cpu-baseline
run --storage rgba8 --no-also
run --math h --no-also
run --math exact --no-also
run --streams 1 --no-also
run --pipeline fused --no-also
run --workload 4k_to_8k_x16 --pipeline fused --no-cpu-baseline --no-also --steps 10 --warmup 2
run --workload 1440p_to_4k_x8 --no-cpu-baseline --no-also --steps 20 --warmup 2
run --rotate-intermediary --no-cpu-baseline --no-also
run --no-fast-paths --no-cpu-baseline --no-also
cat $O/modes.log
