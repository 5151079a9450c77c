#!/bin/bash
# RCAS read-amplification attribution (timing experiments: the no-halo variants compute wrong strip edges).
#   base  = the tree;  nocol = no apron-column loads (-DFSR1_RCAS_NO_HALO);  norow = the apron rows above / below a strip
#   replaced by the strip's own first / last row;  none = both.  Time per launch + FETCH_SIZE / TCC counters per variant.
set -u
ROOT=$PWD; OUT=$ROOT/gpurun_out; export TMPDIR=/tmp
L=variants/libfsr1_
[ -n "${SKIP_AB:-}" ] || python tools/abtest.py --libs ${L}base.so,${L}nocol.so,${L}norow.so,${L}none.so --workloads 1080p_to_4k,1440p_to_4k_x8 --kernels rcas,pair --reps 3 > $OUT/r2c12_ab.log 2>&1
[ -n "${SKIP_AB:-}" ] || cat $OUT/r2c12_ab.log
cd /tmp
for v in base nocol norow none; do
  for pass in "fetch:FETCH_SIZE" "tcc:TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
    name=${pass%%:*}; ctrs=${pass#*:}
    rm -rf /tmp/pmc_${v}_$name
    FSR1_HIP_LIB=$ROOT/${L}$v.so rocprofv3 --kernel-trace --output-format csv --pmc $ctrs -d /tmp/pmc_${v}_$name -o r -- \
      python $ROOT/bench.py --no-cpu-baseline --steps 16 --warmup 4 > /tmp/pmc_${v}_$name.log 2>&1
  done
  python - $v <<'PY' >> $OUT/r2c12_pmc.log
import sys, json
sys.path.insert(0, "/root/repo/tools")
import prof_summary as ps
v = sys.argv[1]
row = {"variant": v}
for name in ("fetch", "tcc"):
    for k, cs in ps.read_pmc("/tmp/pmc_%s_%s" % (v, name)).items():
        if k.startswith("fsr1::rcas_kernel"):
            row.update({c: round(x, 1) for c, x in cs.items()})
print(json.dumps(row))
PY
done
cat $OUT/r2c12_pmc.log
