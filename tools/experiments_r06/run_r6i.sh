#!/bin/bash
# round 6, late: (1) who takes the F-strict re-evaluation pass and at what priority; (2) stream count of the pipelined headline under F-strict
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out
L=gpurun_out/r6i1_strict_pass_wave.log
python tools/abtest.py --libs variants/libfsr1_tree.so@0x40,variants/libfsr1_strict_rot.so@0x40,variants/libfsr1_strict_prio.so@0x40,variants/libfsr1_strict_prio_rot.so@0x40 \
  --workloads 1080p_to_4k,1440p_to_4k --kernels easu,pair,fused --reps 3 > $L 2>&1
L2=gpurun_out/r6i2_strict_streams.log
: > $L2
for rep in 1 2; do for s in 2 3 4; do
  python bench.py --streams $s --steps 1000 --warmup 100 --no-cpu-baseline --no-also --no-latency --no-parity --no-submit-ceiling --no-telemetry-window --no-cold-rcas --no-steady 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(json.dumps({'streams':d['streams'],'value':d['value'],'ms_per_step':d['ms_per_step'],'one_stream':d['config']['one_stream']['value']}))" >> $L2
done; done
cat $L $L2
