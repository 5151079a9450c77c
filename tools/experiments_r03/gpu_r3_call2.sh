#!/bin/bash
# Round 3, GPU call 2: -m gpu suite on the tree with the row-terms table and the pitched LDS layout; A/B of the generic EASU
# (round-2 kernels / row terms only / row terms + pitched layout), stacked RCAS strips on a cold image, fused quad-kernel tile
# heights; PMC + power row of the matrix-pipe EASU variant.
set -u
ROOT=$PWD; OUT=$ROOT/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
L=variants/libfsr1_
timeout 1500 python -m pytest tests -x -q -m gpu --durations=5 > $OUT/r3c2_pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/r3c2_pytest.log
tail -12 $OUT/r3c2_pytest.log
timeout 900 python tools/abtest.py --libs ${L}r2base.so,${L}nopitch.so,@0 --workloads 1440p_to_4k,1270p_to_4k,1662p_to_4k,1440p_to_4k_x8,720p_to_1440p --kernels easu,pair,fused --reps 3 > $OUT/r3c2_ab_generic.log 2>&1
cat $OUT/r3c2_ab_generic.log
timeout 600 python tools/abtest.py --libs @0,${L}rcas_stack2.so,${L}rcas_stack4.so,${L}rcas_stack8.so --workloads 1080p_to_4k,4k_to_8k_x16 --kernels rcas,rcas_cold,pair --reps 3 > $OUT/r3c2_ab_rcas_stack.log 2>&1
cat $OUT/r3c2_ab_rcas_stack.log
timeout 600 python tools/abtest.py --libs @0,${L}fs2qh16.so,${L}fs2qh24.so --workloads 1080p_to_4k,4k_to_8k_x16,540p_to_1080p --kernels fused --reps 3 > $OUT/r3c2_ab_fused_qh.log 2>&1
cat $OUT/r3c2_ab_fused_qh.log
timeout 300 python tools/experiments_r03/power_probe.py --libs ${L}r2base.so@0,${L}mfma_default.so@0 --kernel easu --seconds 6 > $OUT/r3c2_power.log 2>&1
cat $OUT/r3c2_power.log
cd /tmp
for v in r2base mfma_default; do
  for pass in "sq1:SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "sq2:SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "sq3:SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE"; do
    name=${pass%%:*}; ctrs=${pass#*:}
    rm -rf /tmp/pmc_${v}_$name
    FSR1_HIP_LIB=$ROOT/${L}$v.so timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc $ctrs -d /tmp/pmc_${v}_$name -o r -- \
      python $ROOT/bench.py --no-cpu-baseline --no-cold-rcas --pipeline easu --steps 16 --warmup 4 > /tmp/pmc_${v}_$name.log 2>&1 || tail -3 /tmp/pmc_${v}_$name.log
  done
  python - $v <<'PY' >> $OUT/r3c2_pmc_mfma.log
import sys, json
sys.path.insert(0, "/root/repo/tools")
import prof_summary as ps
v = sys.argv[1]
for name in ("sq1", "sq2", "sq3"):
    try:
        for k, cs in ps.read_pmc("/tmp/pmc_%s_%s" % (v, name)).items():
            if "easu" in k:
                print(json.dumps({"variant": v, "pass": name, "kernel": k[:80], **{c: round(x, 1) for c, x in cs.items()}}))
    except Exception as e:
        print(json.dumps({"variant": v, "pass": name, "error": str(e)[:200]}))
PY
done
cat $OUT/r3c2_pmc_mfma.log
