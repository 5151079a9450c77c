#!/bin/bash
# Round 3, GPU call 15: EXACT variants with the exhaustively verified 6-instruction reciprocal (rcp_ieee) instead of the general
# IEEE division: the self-test over all 2^32 operands, EXACT parity tests, A/B against the tree before.
set -u
ROOT=$PWD; OUT=$ROOT/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
L=variants/libfsr1_
python - <<'PY' > $OUT/r3c15_selftest.log 2>&1
import importlib, time
fsr = importlib.import_module("fidelityfx-fsr_amd")
t = time.time(); n = fsr.selftest(); print("selftest failures:", n, "in %.2f s" % (time.time() - t))
PY
cat $OUT/r3c15_selftest.log
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_special_values.py tests/test_gpu_color.py tests/test_gpu_unorm.py "tests/test_gpu_fullframe.py::test_whole_frame_two_pass_and_fused" tests/test_gpu_parity_h.py -x -q -m gpu > $OUT/r3c15_pytest.log 2>&1; echo "rc=$?" >> $OUT/r3c15_pytest.log
tail -4 $OUT/r3c15_pytest.log
timeout 600 python tools/abtest.py --libs ${L}prev.so,@0 --workloads 1080p_to_4k,1440p_to_4k,4k_to_8k_x16 --kernels easu,rcas,pair,fused --math exact --reps 3 > $OUT/r3c15_ab_exact.log 2>&1
cat $OUT/r3c15_ab_exact.log
