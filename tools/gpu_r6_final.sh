#!/bin/bash
# Round 6 closing run on the GPU box: one rocprofv3 stats + PMC summary per BASELINE config and per true-ratio preset at the round's default
# arithmetic (F-strict), plus the default arithmetic's exact-2x pair for comparison (profiles/r06_*; --streams 1 so that a kernel's duration in the
# trace is its own); the default bench line and the driver's K = 20 line, the presets' lines, the default arithmetic's line; the C runner
# (throughput + single-frame latency); the pipelined timeline.  The -m gpu suite runs in its own call (tools/experiments_r06: gpu_pytest).
#   usage (through gpurun): bash tools/gpu_r6_final.sh [profiles|bench|all]
WHAT=${1:-all}
mkdir -p gpurun_out/profiles
P="--streams 1"
if [ "$WHAT" = profiles ] || [ "$WHAT" = all ]; then
NOTE="Round 6, F-strict (the headline arithmetic), one in-order stream" tools/gpu_profile.sh r06_1080p_to_4k_two-pass $P
NOTE="Round 6, the default (F) arithmetic of rounds 1-5, one in-order stream" tools/gpu_profile.sh r06_1080p_to_4k_two-pass_f --math f $P
NOTE="Round 6, BASELINE configs[3], F-strict (one in-order stream: one-step 62 x 14 tiles)" tools/gpu_profile.sh r06_1080p_to_4k_fused --pipeline fused $P
NOTE="Round 6, BASELINE configs[3], default (F) arithmetic (the tall 62 x 30 tile)" tools/gpu_profile.sh r06_1080p_to_4k_fused_f --pipeline fused --math f $P
NOTE="Round 6, BASELINE configs[0] shape: fp32 FsrEasuF, EASU only, RGBA32F storage (F-strict takes the EXACT kernel: no store conversion to test against)" tools/gpu_profile.sh r06_540p_to_1080p_easu_rgba32f --workload 540p_to_1080p --pipeline easu --storage rgba32f $P
NOTE="Round 6, 1.5x 'Quality' single frame, F-strict (generic kernel, 512-thread 64 x 32 tiles)" tools/gpu_profile.sh r06_1440p_to_4k_two-pass --workload 1440p_to_4k $P
NOTE="Round 6, 1.3x 'Ultra Quality' true-ratio preset 2954x1662 -> 4K, F-strict" tools/gpu_profile.sh r06_1662p_to_4k_two-pass --workload 1662p_to_4k $P
NOTE="Round 6, 1.7x 'Balanced' true-ratio preset 2259x1270 -> 4K, F-strict" tools/gpu_profile.sh r06_1270p_to_4k_two-pass --workload 1270p_to_4k $P
STEPS=100 NOTE="Round 6, BASELINE configs[2] per-GPU shard (8 frames per launch), F-strict" tools/gpu_profile.sh r06_1440p_to_4k_x8_two-pass --workload 1440p_to_4k_x8 $P
STEPS=40 PMC_STEPS=6 NOTE="Round 6, BASELINE configs[4] per-GPU shard as ONE fused launch, F-strict" tools/gpu_profile.sh r06_4k_to_8k_x16_fused --workload 4k_to_8k_x16 --pipeline fused $P
fi
if [ "$WHAT" = bench ] || [ "$WHAT" = all ]; then
python bench.py > gpurun_out/r06_bench_default.json 2> gpurun_out/r06_bench_default.err
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06_bench_k20.json 2> gpurun_out/r06_bench_k20.err
python bench.py --math f --no-cpu-baseline > gpurun_out/r06_bench_default_f.json 2> gpurun_out/r06_bench_default_f.err
python bench.py --workload 1440p_to_4k --no-cpu-baseline --no-also > gpurun_out/r06_bench_1440p.json 2> gpurun_out/r06_bench_1440p.err
python bench.py --workload 1662p_to_4k --no-cpu-baseline --no-also > gpurun_out/r06_bench_1662p.json 2> gpurun_out/r06_bench_1662p.err
python bench.py --workload 1270p_to_4k --no-cpu-baseline --no-also > gpurun_out/r06_bench_1270p.json 2> gpurun_out/r06_bench_1270p.err
python bench.py --workload 1440p_to_4k_x8 --no-cpu-baseline --no-also > gpurun_out/r06_bench_1440p_x8.json 2> gpurun_out/r06_bench_1440p_x8.err
python bench.py --workload 4k_to_8k_x16 --pipeline fused --no-cpu-baseline --no-also --steps 60 --warmup 6 > gpurun_out/r06_bench_8k_x16_fused.json 2> gpurun_out/r06_bench_8k_x16_fused.err
python bench.py --submit-only > gpurun_out/r06_submit_1rank.json 2>/dev/null
python bench.py --submit-only --gpus 8 --backend gloo --oversubscribe 2>/dev/null | grep '^{' > gpurun_out/r06_submit_8ranks_one_gpu_gloo.json
for f in gpurun_out/r06_bench_default.json gpurun_out/r06_bench_k20.json gpurun_out/r06_bench_default_f.json; do cut -c1-220 $f; done
R=runner/fsr1_runner; O=gpurun_out/r06_runner_c_host.log; : > $O
for M in strict f; do for S in 3 1; do for PL in two-pass auto; do
  echo "# $R --steps 500 --streams $S --math $M --pipeline $PL --latency 300" >> $O
  timeout 120 $R --steps 500 --streams $S --math $M --pipeline $PL --latency 300 2>/dev/null | grep '^{' >> $O
done; done; done
echo "# $R --gpus 1 --frames 8 --in 2560x1440 --out 3840x2160 --steps 100 --math strict" >> $O
timeout 120 $R --gpus 1 --frames 8 --in 2560x1440 --out 3840x2160 --steps 100 --math strict 2>/dev/null | grep '^{' >> $O
echo "# $R --gpus 1 --frames 16 --in 3840x2160 --out 7680x4320 --steps 20 --pipeline auto --math strict" >> $O
timeout 200 $R --gpus 1 --frames 16 --in 3840x2160 --out 7680x4320 --steps 20 --pipeline auto --math strict 2>/dev/null | grep '^{' >> $O
cut -c1-160 $O
fi
