#!/bin/bash
# Builds the two diagnostic variants of the F-strict kernels from the CURRENT tree (wrong results on purpose, timing only):
#   variants/libfsr1_strict_nodrain.so  the queued pixels are never re-evaluated (detection + push cost)
#   variants/libfsr1_strict_detonly.so  nothing is pushed either (detection cost)
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
W=/tmp/fsr1_breakdown; rm -rf $W
for d in a b c; do mkdir -p $W/$d/include $W/$d/fidelityfx-fsr_amd/csrc; cp $ROOT/include/fsr1_device_easu.hpp $W/$d/include/; done
python3 - $W <<'PY'
import sys
w=sys.argv[1]
for d in ('b','c'):
    p='%s/%s/include/fsr1_device_easu.hpp'%(w,d)
    s=open(p).read()
    a="    for (int i = tid; i < n; i += THREADS) redo((int)q.ids[i]);"
    assert a in s
    s=s.replace(a,"    for (int i = tid; i < 0 * n; i += THREADS) redo((int)q.ids[i]);")
    if d=='c':
        a="  if (mask) {\n    uint32_t at = atomicAdd(q.count"
        assert a in s
        s=s.replace(a,"  if (mask == 0xffffffffu) {\n    uint32_t at = atomicAdd(q.count")
    open(p,'w').write(s)
PY
(cd $W && diff -ru a b > $ROOT/tools/experiments_r06/strict_nodrain.patch || true; mv b bb; mv c b; diff -ru a b > $ROOT/tools/experiments_r06/strict_detect_only.patch || true)
cd $ROOT && tools/build_variant.sh strict_nodrain "" tools/experiments_r06/strict_nodrain.patch && tools/build_variant.sh strict_detonly "" tools/experiments_r06/strict_detect_only.patch
