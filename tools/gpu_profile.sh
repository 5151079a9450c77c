#!/bin/bash
# Runs on the GPU box (through gpurun): kernel-trace stats + separate PMC passes of the bench command.
#   tools/gpu_profile.sh TAG [bench args...]   -> gpurun_out/prof_TAG/{stats,fetch,write,tcc,sq1,sq2}
# PMC passes are collected in their own runs, never combined with tracing domains other than kernel-trace.
set -u
TAG=$1; shift
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
BENCH="python $PWD/bench.py --no-cpu-baseline $*"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o r -- $BENCH --steps 400 --warmup 50 > "$OUT/stats.log" 2>&1
for pass in "fetch:FETCH_SIZE" "write:WRITE_SIZE" "tcc:TCC_HIT_sum TCC_MISS_sum" \
            "sq1:SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD" \
            "sq2:SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_WR"; do
  name=${pass%%:*}; ctrs=${pass#*:}
  rocprofv3 --kernel-trace --output-format csv --pmc $ctrs -d "$OUT/$name" -o r -- $BENCH --steps 20 --warmup 4 > "$OUT/$name.log" 2>&1
done
grep -h '^{' "$OUT/stats.log" | tail -1
