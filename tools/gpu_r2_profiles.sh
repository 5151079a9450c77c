#!/bin/bash
# round-2 profiles: one rocprofv3 kernel-trace + PMC summary per BASELINE config (tools/gpu_profile.sh), then the default bench line
mkdir -p gpurun_out/profiles
R=${ROUND_TAG:-r02}
NOTE="Round 2" tools/gpu_profile.sh ${R}_1080p_to_4k_two-pass
NOTE="Round 2, BASELINE configs[3]" tools/gpu_profile.sh ${R}_1080p_to_4k_fused --pipeline fused
NOTE="Round 2, packed-fp16 entry points (FsrEasuH / FsrRcasH)" tools/gpu_profile.sh ${R}_1080p_to_4k_two-pass_h --math h
NOTE="Round 2, BASELINE configs[0] shape: fp32 FsrEasuF, EASU only, RGBA32F storage" tools/gpu_profile.sh ${R}_540p_to_1080p_easu_rgba32f --workload 540p_to_1080p --pipeline easu --storage rgba32f
NOTE="Round 2, 1.5x single frame" tools/gpu_profile.sh ${R}_1440p_to_4k_two-pass --workload 1440p_to_4k
STEPS=100 NOTE="Round 2, BASELINE configs[2] per-GPU shard (8 frames per launch)" tools/gpu_profile.sh ${R}_1440p_to_4k_x8_two-pass --workload 1440p_to_4k_x8
STEPS=40 PMC_STEPS=6 NOTE="Round 2, BASELINE configs[4] per-GPU shard (16 frames per launch)" tools/gpu_profile.sh ${R}_4k_to_8k_x16_two-pass --workload 4k_to_8k_x16
STEPS=40 PMC_STEPS=6 NOTE="Round 2, BASELINE configs[4] per-GPU shard as ONE fused launch (what auto runs at exactly 2x)" tools/gpu_profile.sh ${R}_4k_to_8k_x16_fused --workload 4k_to_8k_x16 --pipeline fused
python bench.py > gpurun_out/${R}_bench_default.json 2> gpurun_out/${R}_bench_default.err
tail -c 600 gpurun_out/${R}_bench_default.json
ls -la gpurun_out/profiles | head -40
