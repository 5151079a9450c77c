#!/bin/bash
# Round 5, GPU call 4: RCAS 32-row strips for batches; packed-fp16 fused vs two dispatches (the auto rule for H); the pipelined timeline.
O=gpurun_out/r5c4; mkdir -p $O
export TMPDIR=/tmp
python tools/abtest.py --libs fidelityfx-fsr_amd/libfsr1_hip.so,variants/libfsr1_rcas32.so --workloads 1440p_to_4k_x8,4k_to_8k_x16,1080p_to_4k_x4 --kernels rcas,pair --reps 3 --launches 80 > $O/ab_rcas32.log 2>&1; cat $O/ab_rcas32.log
python tools/abtest.py --libs fidelityfx-fsr_amd/libfsr1_hip.so --workloads 1080p_to_4k,1080p_to_4k_x4,4k_to_8k,4k_to_8k_x16,540p_to_1080p,720p_to_1440p --kernels pair,fused --math h --reps 2 --launches 100 > $O/ab_h_auto.log 2>&1; cat $O/ab_h_auto.log
cd /tmp
for cfg in "two-pass:3" "two-pass:1" "fused:3" "fused:1"; do
  pl=${cfg%%:*}; st=${cfg#*:}
  rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_${pl}_$st -o r -- python $OLDPWD/bench.py --no-cpu-baseline --no-cold-rcas --no-also --no-latency --no-steady --no-parity --pipeline $pl --streams $st --steps 300 --warmup 50 --regions 3 > /tmp/tl_${pl}_$st.log 2>&1
  echo "=== $pl, $st stream(s)" >> $OLDPWD/$O/pipeline_timeline.log
  python $OLDPWD/tools/pipeline_timeline.py /tmp/tl_${pl}_$st --skip-first 800 --last 1200 >> $OLDPWD/$O/pipeline_timeline.log 2>&1
done
cd $OLDPWD; cat $O/pipeline_timeline.log
