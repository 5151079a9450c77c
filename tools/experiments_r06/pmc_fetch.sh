#!/bin/bash
# usage (GPU box): tools/experiments_r06/pmc_fetch.sh TAG LIB [bench args]  -> per-kernel FETCH_SIZE / WRITE_SIZE / TCC hit, miss of one bench configuration
# (separate --pmc passes, kernel-trace only; FETCH_SIZE x 2 per the guide's gfx950 correction)
TAG=$1; LIB=$2; shift; shift
ROOT=$PWD; RAW=/tmp/pmc_$TAG; rm -rf $RAW; mkdir -p $RAW; export TMPDIR=/tmp
[ -n "$LIB" ] && export FSR1_HIP_LIB=$ROOT/$LIB
BENCH="python $ROOT/bench.py --no-cpu-baseline --no-cold-rcas --no-also --no-latency --no-steady --no-parity --no-telemetry-window --no-submit-ceiling --streams 1 $*"
cd /tmp
for pass in "fetch:FETCH_SIZE" "write:WRITE_SIZE" "tcc:TCC_HIT_sum TCC_MISS_sum"; do
  name=${pass%%:*}; ctrs=${pass#*:}
  rocprofv3 --kernel-trace --output-format csv --pmc $ctrs -d $RAW/$name -o r -- $BENCH --steps 16 --warmup 4 > $RAW/$name.log 2>&1
done
cd $ROOT
python - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$RAW/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if "fsr1::" in k:
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in agg.items():
    row = {n: sum(v) / len(v) for n, v in c.items()}
    out = {"tag": "$TAG", "kernel": k}
    if "FETCH_SIZE" in row: out["fetch_MB"] = round(row["FETCH_SIZE"] * 2 * 1024 / 1e6, 2)
    if "WRITE_SIZE" in row: out["write_MB"] = round(row["WRITE_SIZE"] * 1024 / 1e6, 2)
    if "TCC_HIT_sum" in row: out["tcc_hit"] = round(row["TCC_HIT_sum"]); out["tcc_miss"] = round(row["TCC_MISS_sum"])
    print(out)
PY
