#!/bin/bash
# usage: kstat.sh <csrc dir> <file.hip> [name regex] [extra flags]  -> per-kernel vgpr/sgpr + instruction counts from the gfx950 assembly
D=$1; SRC=$2; F=${3:-.}; shift 3
cd $D
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fno-slp-vectorize -I../../include -I. --cuda-device-only -S -o /tmp/kstat_$$.s $SRC "$@" 2>/dev/null
python3 - /tmp/kstat_$$.s "$F" <<'PY'
import re,sys,subprocess
txt=open(sys.argv[1]).read(); F=sys.argv[2]
# split on kernel labels
parts=re.split(r"\n(_Z[\w]+):[^\n]*\n",txt)
stats={}
for i in range(1,len(parts),2):
    name,body=parts[i],parts[i+1]
    body=body.split('.Lfunc_end')[0]
    ins=[l.strip() for l in body.split('\n') if re.match(r'\s+[a-z]+_[a-z0-9_]+',l)]
    def c(p): return sum(1 for l in ins if re.match(p,l))
    stats[name]=(len(ins),c(r'v_'),c(r'v_(fma|mul|add|sub|fmac|mac)_f32'),c(r'v_pk_'),c(r'ds_'),c(r'(global|buffer|flat)_'),c(r's_waitcnt'),c(r'v_(rcp|rsq|sqrt)'))
meta={}
for m in re.finditer(r'\.name:\s+(\S+)\n(.*?)\.wavefront_size',txt,re.S):
    pass
for blk in txt.split('- .agpr_count')[1:]:
    n=re.search(r'\.name:\s+(\S+)',blk).group(1)
    meta[n]=(re.search(r'\.vgpr_count:\s+(\d+)',blk).group(1),re.search(r'\.sgpr_count:\s+(\d+)',blk).group(1),re.search(r'\.vgpr_spill_count:\s+(\d+)',blk).group(1))
for n,s in stats.items():
    if n not in meta: continue
    dem=subprocess.run(['c++filt',n],capture_output=True,text=True).stdout.strip()
    dem=re.sub(r'\(.*','',dem)
    if re.search(F,dem):
        print("%-56s vgpr %3s sgpr %3s spill %s | insts %5d valu %5d fp32fma-class %4d pk %3d ds %3d vmem %3d wait %3d trans %2d"%((dem[:56],)+meta[n]+s))
PY
rm -f /tmp/kstat_$$.s
