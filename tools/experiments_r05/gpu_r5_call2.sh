#!/bin/bash
# Round 5, GPU call 2: A/B of three kernel experiments (variants/ built by tools/build_variant.sh from tools/experiments_r05/*.patch)
O=gpurun_out/r5c2; mkdir -p $O
export TMPDIR=/tmp
# parity of the staging variant first (bit-identical by construction: same arithmetic, different thread -> texel mapping)
FSR1_HIP_LIB=$PWD/variants/libfsr1_stagerows.so timeout 600 python -m pytest tests/test_gpu_image_parity.py tests/test_gpu_parity.py -q -x -k "true_ratio or golden or ragged or 1440p" > $O/pytest_stagerows.log 2>&1; echo "rc=$?" >> $O/pytest_stagerows.log; tail -3 $O/pytest_stagerows.log
python tools/abtest.py --libs variants/libfsr1_base.so,variants/libfsr1_stagerows.so --workloads 1440p_to_4k,1270p_to_4k,1662p_to_4k,831p_to_1080p,1440p_to_4k_x8 --kernels easu --reps 3 > $O/ab_stagerows.log 2>&1; cat $O/ab_stagerows.log
python tools/abtest.py --libs variants/libfsr1_base.so,variants/libfsr1_rcasnt.so --workloads 1080p_to_4k,1440p_to_4k_x8,4k_to_8k_x16 --kernels rcas,rcas_cold,pair --reps 3 > $O/ab_rcasnt.log 2>&1; cat $O/ab_rcasnt.log
python tools/abtest.py --libs variants/libfsr1_base.so,variants/libfsr1_base.so@0x400,variants/libfsr1_fusededge.so --workloads 1080p_to_4k,4k_to_8k --kernels fused --reps 3 > $O/ab_fused_stores.log 2>&1; cat $O/ab_fused_stores.log
cd /tmp
for v in base:0 base:0x400 fusededge:0; do
  lib=${v%%:*}; fl=${v#*:}
  for c in WRITE_SIZE FETCH_SIZE; do
    FSR1_HIP_LIB=$OLDPWD/variants/libfsr1_$lib.so FSR1_AB_FLAGS=$fl rocprofv3 --kernel-trace --output-format csv --pmc $c -d /tmp/pmc_${lib}_${fl}_$c -o r -- python $OLDPWD/tools/abtest.py --child --workloads 1080p_to_4k --kernels fused --launches 30 --ramp 0.05 > /tmp/pmc_${lib}_${fl}_$c.log 2>&1
    echo "$lib flags=$fl: $(python $OLDPWD/tools/experiments_r05/pmc_avg.py /tmp/pmc_${lib}_${fl}_$c fused_s2)" >> $OLDPWD/$O/pmc_fused_stores.log
  done
done
cd $OLDPWD; cat $O/pmc_fused_stores.log
