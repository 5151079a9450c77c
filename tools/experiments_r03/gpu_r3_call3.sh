#!/bin/bash
# Round 3, GPU call 3: generic EASU with permuted lane columns (no LDS bank conflicts), bounds from the taps, a luma plane for
# phase 2 and the pitched layout — tests, A/B against the round-2 kernels, LDS counters of the generic kernel.
set -u
ROOT=$PWD; OUT=$ROOT/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
L=variants/libfsr1_
timeout 1500 python -m pytest tests -x -q -m gpu --durations=5 > $OUT/r3c3_pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/r3c3_pytest.log
tail -12 $OUT/r3c3_pytest.log
timeout 900 python tools/abtest.py --libs ${L}r2base.so,@0 --workloads 1440p_to_4k,1270p_to_4k,1662p_to_4k,1440p_to_4k_x8,720p_to_1440p,1080p_to_4k,4k_to_8k_x16 --kernels easu,pair,fused --reps 3 > $OUT/r3c3_ab_generic.log 2>&1
cat $OUT/r3c3_ab_generic.log
cd /tmp
for v in r2base tree; do
  lib=$ROOT/${L}$v.so; [ $v = tree ] && lib=$ROOT/fidelityfx-fsr_amd/libfsr1_hip.so
  for pass in "sq1:SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "sq2:SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"; do
    name=${pass%%:*}; ctrs=${pass#*:}
    rm -rf /tmp/pmc_${v}_$name
    FSR1_HIP_LIB=$lib timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc $ctrs -d /tmp/pmc_${v}_$name -o r -- \
      python $ROOT/bench.py --no-cpu-baseline --no-cold-rcas --workload 1440p_to_4k --pipeline easu --steps 16 --warmup 4 > /tmp/pmc_${v}_$name.log 2>&1 || tail -3 /tmp/pmc_${v}_$name.log
  done
  python - $v <<'PY' >> $OUT/r3c3_pmc_generic.log
import sys, json
sys.path.insert(0, "/root/repo/tools")
import prof_summary as ps
v = sys.argv[1]
for name in ("sq1", "sq2"):
    try:
        for k, cs in ps.read_pmc("/tmp/pmc_%s_%s" % (v, name)).items():
            if "easu" in k:
                print(json.dumps({"variant": v, "pass": name, "kernel": k[:80], **{c: round(x, 1) for c, x in cs.items()}}))
    except Exception as e:
        print(json.dumps({"variant": v, "pass": name, "error": str(e)[:200]}))
PY
done
cat $OUT/r3c3_pmc_generic.log
