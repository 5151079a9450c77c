#!/bin/bash
# Round 5 closing run on the GPU box: the whole -m gpu suite; one rocprofv3 stats + PMC summary per BASELINE config AND per true-ratio
# preset of the reference's sample (profiles/r05_*; --streams 1 so that a kernel's duration in the trace is its own); the pipelined
# timelines; the default bench line and the driver's K = 20 line; the C runner; the hand copy line.
export ROUND_TAG=r05
mkdir -p gpurun_out/profiles
timeout 1500 python -m pytest tests -m gpu -q --durations=6 > gpurun_out/r05_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r05_pytest.log
tail -4 gpurun_out/r05_pytest.log
P="--streams 1"
NOTE="Round 5 (one in-order stream)" tools/gpu_profile.sh r05_1080p_to_4k_two-pass $P
NOTE="Round 5, BASELINE configs[3] (one in-order stream: the tall 62 x 30 tile)" tools/gpu_profile.sh r05_1080p_to_4k_fused --pipeline fused $P
NOTE="Round 5, packed-fp16 entry points (FsrEasuH / FsrRcasH)" tools/gpu_profile.sh r05_1080p_to_4k_two-pass_h --math h $P
NOTE="Round 5, BASELINE configs[0] shape: fp32 FsrEasuF, EASU only, RGBA32F storage" tools/gpu_profile.sh r05_540p_to_1080p_easu_rgba32f --workload 540p_to_1080p --pipeline easu --storage rgba32f $P
NOTE="Round 5, 1.5x 'Quality' single frame (generic kernel, 512-thread 64 x 32 tiles)" tools/gpu_profile.sh r05_1440p_to_4k_two-pass --workload 1440p_to_4k $P
NOTE="Round 5, 1.3x 'Ultra Quality' true-ratio preset 2954x1662 -> 4K (FSRSample.h:79-95)" tools/gpu_profile.sh r05_1662p_to_4k_two-pass --workload 1662p_to_4k $P
NOTE="Round 5, 1.7x 'Balanced' true-ratio preset 2259x1270 -> 4K" tools/gpu_profile.sh r05_1270p_to_4k_two-pass --workload 1270p_to_4k $P
NOTE="Round 5, 1.3x 'Ultra Quality' true-ratio preset 1477x831 -> 1080p" tools/gpu_profile.sh r05_831p_to_1080p_two-pass --workload 831p_to_1080p $P
STEPS=100 NOTE="Round 5, BASELINE configs[2] per-GPU shard (8 frames per launch)" tools/gpu_profile.sh r05_1440p_to_4k_x8_two-pass --workload 1440p_to_4k_x8 $P
STEPS=40 PMC_STEPS=6 NOTE="Round 5, BASELINE configs[4] per-GPU shard as ONE fused launch (what auto runs at exactly 2x)" tools/gpu_profile.sh r05_4k_to_8k_x16_fused --workload 4k_to_8k_x16 --pipeline fused $P
# the pipelined regime (three streams, the default): kernel-trace only — durations include the overlap with the neighbouring frames;
# tools/pipeline_timeline.py turns the trace into residency / chip-time-per-frame figures
cd /tmp && export TMPDIR=/tmp
for cfg in "two-pass:3:1080p_to_4k" "two-pass:1:1080p_to_4k" "fused:3:1080p_to_4k" "two-pass:3:1440p_to_4k"; do
  pl=$(echo $cfg | cut -d: -f1); st=$(echo $cfg | cut -d: -f2); wl=$(echo $cfg | cut -d: -f3)
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tl_${wl}_${pl}_$st -o r -- python $OLDPWD/bench.py --no-cpu-baseline --no-cold-rcas --no-also --no-latency --no-steady --no-parity --workload $wl --pipeline $pl --streams $st --steps 300 --warmup 50 --regions 3 > /tmp/tl_${wl}_${pl}_$st.log 2>&1
  echo "=== $wl, $pl, $st stream(s)" >> $OLDPWD/gpurun_out/profiles/r05_pipeline_timeline.log
  python $OLDPWD/tools/pipeline_timeline.py /tmp/tl_${wl}_${pl}_$st --skip-first 800 --last 1200 >> $OLDPWD/gpurun_out/profiles/r05_pipeline_timeline.log 2>&1
  if [ "$st" = "3" ]; then find /tmp/tl_${wl}_${pl}_$st -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} $OLDPWD/gpurun_out/profiles/r05_${wl}_${pl}_pipelined_kernel_stats.csv; fi
done
cd $OLDPWD
cat gpurun_out/profiles/r05_pipeline_timeline.log | grep -E "===|window|chip time"
python bench.py > gpurun_out/r05_bench_default.json 2> gpurun_out/r05_bench_default.err
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05_bench_k20.json 2> gpurun_out/r05_bench_k20.err
python bench.py --workload 1440p_to_4k --no-cpu-baseline > gpurun_out/r05_bench_1440p.json 2> gpurun_out/r05_bench_1440p.err
python bench.py --workload 1662p_to_4k --no-cpu-baseline --no-also > gpurun_out/r05_bench_1662p.json 2> gpurun_out/r05_bench_1662p.err
python bench.py --workload 1270p_to_4k --no-cpu-baseline --no-also > gpurun_out/r05_bench_1270p.json 2> gpurun_out/r05_bench_1270p.err
cut -c1-200 gpurun_out/r05_bench_default.json; cut -c1-200 gpurun_out/r05_bench_k20.json
R=runner/fsr1_runner; O=gpurun_out/r05_runner_c_host.log; : > $O
for S in 3 1; do
timeout 300 $R --gpus 1 --steps 2000 --warmup 100 --streams $S >> $O 2>/dev/null
timeout 300 $R --gpus 1 --steps 2000 --warmup 100 --pipeline auto --streams $S >> $O 2>/dev/null
timeout 300 $R --gpus 1 --in 2560x1440 --out 3840x2160 --steps 1000 --warmup 100 --streams $S >> $O 2>/dev/null
timeout 300 $R --gpus 1 --frames 8 --in 2560x1440 --out 3840x2160 --steps 100 --warmup 10 --streams $S >> $O 2>/dev/null
timeout 300 $R --gpus 1 --frames 16 --in 3840x2160 --out 7680x4320 --steps 30 --warmup 5 --pipeline auto --streams $S >> $O 2>/dev/null
done
grep '^{' $O | cut -c1-200
timeout 300 tools/ubench/copy_bench > gpurun_out/r05_copy_bench.log 2>&1
ls gpurun_out/profiles | head -60
