#!/bin/bash
# Runs on the GPU box (through gpurun): kernel-trace stats + separate PMC passes of one bench.py configuration, digested
# into profiles-style summaries.  Raw rocprofv3 output stays in /tmp on the box; only the summaries come back.
#   tools/gpu_profile.sh TAG [bench args...]   -> gpurun_out/profiles/TAG.{json,md} + TAG_kernel_stats.csv
# (the bench line of the traced run is an INPUT of the summary — its stopwatch sits beside the trace's averages in TAG.json — and is not kept
#  as a file of its own: written before the summary existed, round 5's TAG.line files disclaimed the traffic of the JSON next to them)
# PMC passes are collected in their own runs, never combined with tracing domains other than kernel-trace.
# Env: STEPS (kernel-trace steps, default 300), PMC_STEPS (default 16)
set -u
TAG=$1; shift
ROOT=$PWD
OUT=$ROOT/gpurun_out/profiles
RAW=/tmp/prof_$TAG
rm -rf "$RAW"; mkdir -p "$OUT" "$RAW"
export TMPDIR=/tmp
STEPS=${STEPS:-300}; PMC_STEPS=${PMC_STEPS:-16}
BENCH="python $ROOT/bench.py --no-cpu-baseline --no-cold-rcas --no-also --no-latency --no-steady --no-parity --no-telemetry-window --no-submit-ceiling $*"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$RAW/stats" -o r -- $BENCH --steps $STEPS --warmup 30 > "$RAW/stats.log" 2>&1
grep -h '^{' "$RAW/stats.log" | tail -1 > "$RAW/$TAG.line"
for pass in "fetch:FETCH_SIZE" "write:WRITE_SIZE" "tcc:TCC_HIT_sum TCC_MISS_sum" \
            "sq1:SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD" \
            "sq2:SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_WR"; do
  name=${pass%%:*}; ctrs=${pass#*:}
  rocprofv3 --kernel-trace --output-format csv --pmc $ctrs -d "$RAW/$name" -o r -- $BENCH --steps $PMC_STEPS --warmup 4 > "$RAW/$name.log" 2>&1
done
cd "$ROOT"
WL=1080p_to_4k; PL=two-pass; MATH=strict; ST=rgba16f  # bench.py's defaults
args=("$@")
for ((i=0;i<${#args[@]};i++)); do
  case "${args[$i]}" in --workload) WL=${args[$((i+1))]};; --pipeline) PL=${args[$((i+1))]};; --math) MATH=${args[$((i+1))]};; --storage) ST=${args[$((i+1))]};; esac
done
python tools/prof_summary.py --out "$OUT/$TAG" --stats "$RAW/stats" --pmc "$RAW/fetch" --pmc "$RAW/write" --pmc "$RAW/tcc" --pmc "$RAW/sq1" --pmc "$RAW/sq2" \
  --workload "$WL" --pipeline "$PL" --math "$MATH" --storage "$ST" --lib "$ROOT/fidelityfx-fsr_amd/libfsr1_hip.so" --bench-line "$RAW/$TAG.line" \
  --command "tools/gpu_profile.sh $TAG $*  (rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline --no-cold-rcas --no-also --no-latency --no-steady --no-parity --no-telemetry-window --no-submit-ceiling $* --steps $STEPS --warmup 30; PMC in separate --pmc passes of $((PMC_STEPS+4)) steps)" \
  --note "${NOTE:-}" > "$RAW/summary.log" 2>&1 || cat "$RAW/summary.log"
find "$RAW/stats" -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} "$OUT/${TAG}_kernel_stats.csv"
tail -2 "$RAW/summary.log"; cut -c1-300 "$RAW/$TAG.line"
