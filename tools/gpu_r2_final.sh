#!/bin/bash
# full gpu test suite, then the per-config profiles and the default bench line (ROUND_TAG names the outputs)
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q --durations=6 > gpurun_out/${ROUND_TAG:-r02}_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/${ROUND_TAG:-r02}_pytest.log
tail -12 gpurun_out/${ROUND_TAG:-r02}_pytest.log
bash tools/gpu_r2_profiles.sh 2>&1 | grep -v "^wrote" | cut -c1-220
