#!/bin/bash
# Round 4 closing run on the GPU box: the whole -m gpu suite, one rocprofv3 stats + PMC summary per BASELINE config (profiles/r04_*,
# taken with --streams 1 so that a kernel's duration in the trace is its own, not its overlap with a neighbour), the headline's
# trace once more under two-stream pipelining (stats only), the default bench line and the driver's K = 20 line, the C runner.
export ROUND_TAG=r04
mkdir -p gpurun_out/profiles
timeout 1200 python -m pytest tests -m gpu -q --durations=6 > gpurun_out/r04_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r04_pytest.log
tail -4 gpurun_out/r04_pytest.log
P="--streams 1"
NOTE="Round 4 (one in-order stream)" tools/gpu_profile.sh r04_1080p_to_4k_two-pass $P
NOTE="Round 4, BASELINE configs[3] (one in-order stream: the tall 62 x 30 tile)" tools/gpu_profile.sh r04_1080p_to_4k_fused --pipeline fused $P
NOTE="Round 4, packed-fp16 entry points (FsrEasuH / FsrRcasH)" tools/gpu_profile.sh r04_1080p_to_4k_two-pass_h --math h $P
NOTE="Round 4, BASELINE configs[0] shape: fp32 FsrEasuF, EASU only, RGBA32F storage" tools/gpu_profile.sh r04_540p_to_1080p_easu_rgba32f --workload 540p_to_1080p --pipeline easu --storage rgba32f $P
NOTE="Round 4, 1.5x single frame" tools/gpu_profile.sh r04_1440p_to_4k_two-pass --workload 1440p_to_4k $P
STEPS=100 NOTE="Round 4, BASELINE configs[2] per-GPU shard (8 frames per launch)" tools/gpu_profile.sh r04_1440p_to_4k_x8_two-pass --workload 1440p_to_4k_x8 $P
STEPS=40 PMC_STEPS=6 NOTE="Round 4, BASELINE configs[4] per-GPU shard as ONE fused launch (what auto runs at exactly 2x)" tools/gpu_profile.sh r04_4k_to_8k_x16_fused --workload 4k_to_8k_x16 --pipeline fused $P
# the pipelined regime (three streams, the default): kernel-trace stats only — durations include the overlap with the neighbouring frames
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_r04_pipelined -o r -- python $OLDPWD/bench.py --no-cpu-baseline --no-cold-rcas --no-also --steps 300 --warmup 30 > /tmp/prof_r04_pipelined.log 2>&1
cd $OLDPWD
find /tmp/prof_r04_pipelined -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} gpurun_out/profiles/r04_1080p_to_4k_two-pass_pipelined_kernel_stats.csv
grep -h '^{' /tmp/prof_r04_pipelined.log | tail -1 > gpurun_out/profiles/r04_1080p_to_4k_two-pass_pipelined.line
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_r04_pipelined_f -o r -- python bench.py --no-cpu-baseline --no-cold-rcas --no-also --pipeline fused --steps 300 --warmup 30 > /tmp/prof_r04_pipelined_f.log 2>&1
find /tmp/prof_r04_pipelined_f -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} gpurun_out/profiles/r04_1080p_to_4k_fused_pipelined_kernel_stats.csv
grep -h '^{' /tmp/prof_r04_pipelined_f.log | tail -1 > gpurun_out/profiles/r04_1080p_to_4k_fused_pipelined.line
python bench.py > gpurun_out/r04_bench_default.json 2> gpurun_out/r04_bench_default.err
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04_bench_k20.json 2> gpurun_out/r04_bench_k20.err
cut -c1-200 gpurun_out/r04_bench_default.json; cut -c1-200 gpurun_out/r04_bench_k20.json
R=runner/fsr1_runner; O=gpurun_out/r04_runner_c_host.log; : > $O
for S in 3 1; do
timeout 300 $R --gpus 1 --steps 2000 --warmup 100 --streams $S >> $O 2>/dev/null
timeout 300 $R --gpus 1 --steps 2000 --warmup 100 --pipeline auto --streams $S >> $O 2>/dev/null
timeout 300 $R --gpus 1 --steps 2000 --warmup 100 --math h --streams $S >> $O 2>/dev/null
timeout 300 $R --gpus 1 --in 2560x1440 --out 3840x2160 --steps 1000 --warmup 100 --streams $S >> $O 2>/dev/null
timeout 300 $R --gpus 1 --frames 8 --in 2560x1440 --out 3840x2160 --steps 100 --warmup 10 --streams $S >> $O 2>/dev/null
timeout 300 $R --gpus 1 --frames 16 --in 3840x2160 --out 7680x4320 --steps 30 --warmup 5 --pipeline auto --streams $S >> $O 2>/dev/null
done
grep '^{' $O | cut -c1-200
ls gpurun_out/profiles | head -40
