#!/bin/bash
# Round 3 closing run on the GPU box: the whole -m gpu suite, one rocprofv3 stats + PMC summary per BASELINE config
# (profiles/r03_*), the default bench line, and the C runner's figures.
export ROUND_TAG=r03
sed -i 's/Round 2/Round 3/g' tools/gpu_r2_profiles.sh  # (notes in the summaries; the copy on the box only)
bash tools/gpu_r2_final.sh
R=runner/fsr1_runner; O=gpurun_out/r03_runner_c_host.log; : > $O
timeout 300 $R --gpus 1 --steps 2000 --warmup 100 >> $O 2>/dev/null
timeout 300 $R --gpus 1 --steps 2000 --warmup 100 --pipeline auto >> $O 2>/dev/null
timeout 300 $R --gpus 1 --steps 2000 --warmup 100 --pipeline fused >> $O 2>/dev/null
timeout 300 $R --gpus 1 --frames 16 --in 3840x2160 --out 7680x4320 --steps 30 --warmup 5 --pipeline auto >> $O 2>/dev/null
timeout 300 $R --gpus 1 --steps 2000 --warmup 100 --math h >> $O 2>/dev/null
timeout 300 $R --gpus 1 --in 2560x1440 --out 3840x2160 --steps 1000 --warmup 100 >> $O 2>/dev/null
timeout 300 $R --gpus 1 --in 2560x1440 --out 3840x2160 --steps 1000 --warmup 100 --pipeline auto >> $O 2>/dev/null
grep '^{' $O | cut -c1-260
