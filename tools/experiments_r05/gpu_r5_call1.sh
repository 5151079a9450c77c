#!/bin/bash
# Round 5, GPU call 1: the new image-level parity tests and pipeline tests, the hand copy line, telemetry files, the new bench blocks,
# baseline kernel times of the true-ratio presets.
O=gpurun_out/r5c1; mkdir -p $O
( ls /sys/class/drm/; for c in /sys/class/drm/card*/device; do echo "== $c -> $(readlink -f $c)"; cat $c/vendor; ls $c/hwmon/hwmon*/ 2>/dev/null | tr '\n' ' '; echo; for f in $c/hwmon/hwmon*/freq1_input $c/hwmon/hwmon*/power1_average $c/hwmon/hwmon*/power1_input $c/hwmon/hwmon*/power1_cap; do [ -e $f ] && echo "$f = $(cat $f 2>&1)"; done; done
  python - <<'PY'
import torch, time
p = torch.cuda.get_device_properties(0)
print("props:", [(k, getattr(p, k)) for k in dir(p) if "pci" in k.lower()])
PY
) > $O/telemetry_probe.log 2>&1
timeout 900 python -m pytest tests/test_gpu_image_parity.py tests/test_gpu_pipeline.py -q -x -s > $O/pytest_new.log 2>&1; echo "rc=$?" >> $O/pytest_new.log
tail -3 $O/pytest_new.log
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity_h.py tests/test_runner.py tests/test_abi.py -q -x -k "run_steps or tall or ring or steps or roctx" > $O/pytest_hooks.log 2>&1; echo "rc=$?" >> $O/pytest_hooks.log
tail -3 $O/pytest_hooks.log
timeout 300 tools/ubench/copy_bench > $O/copy_bench.log 2>&1; tail -5 $O/copy_bench.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_k20.json 2> $O/bench_k20.err; echo "bench rc=$?"; cut -c1-400 $O/bench_k20.json; tail -3 $O/bench_k20.err
timeout 600 python tools/abtest.py --workloads 1440p_to_4k,1270p_to_4k,1662p_to_4k,831p_to_1080p,1440p_to_4k_x8 --kernels easu,rcas,rcas_cold,pair --reps 2 > $O/presets_baseline.log 2>&1; cat $O/presets_baseline.log
