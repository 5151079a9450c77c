#!/usr/bin/env python3
"""How the per-launch times depend on the number of frame sets the steps rotate over (bench.py: enough sets to exceed 320 MB;
the Infinity Cache holds 256 MB): EASU, RCAS, the pair, and torch's elementwise copy / add of one 4K image for scale."""
import importlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
fsr = importlib.import_module("fidelityfx-fsr_amd"); fsr.load()
dev = torch.device("cuda", 0)
iw, ih, ow, oh = 1920, 1080, 3840, 2160
timer = fsr.Timer()
econ = fsr.FsrEasuCon(iw, ih, iw, ih, ow, oh); rcon = fsr.FsrRcasCon(0.25)
base = torch.from_numpy(fsr.frames.synthetic_frame(iw, ih, k=1)).to(dev)

def us(fn, n=400):
    t0 = time.perf_counter(); i = 0
    while time.perf_counter() - t0 < 0.25:
        fn(i); i += 1
    torch.cuda.synchronize()
    timer.start()
    for i in range(n):
        fn(i)
    timer.stop()
    return round(timer.elapsed_ms() / n * 1e3, 2)

for ring in (1, 2, 3, 4, 6, 12, 24):
    srcs = [torch.roll(base, shifts=(3 * s, 5 * s), dims=(0, 1)).contiguous() for s in range(ring)]
    mids = [torch.empty(oh, ow, 4, dtype=torch.float16, device=dev) for _ in range(ring)]
    dsts = [torch.empty(oh, ow, 4, dtype=torch.float16, device=dev) for _ in range(ring)]
    for s in range(ring):
        fsr.easu(srcs[s], mids[s], con=econ)
    def easu(i): fsr.easu(srcs[i % ring], mids[i % ring], con=econ)
    def rcas(i): fsr.rcas(mids[i % ring], dsts[i % ring], con=rcon)
    def pair(i): easu(i); rcas(i)
    def copy(i): dsts[i % ring].copy_(mids[i % ring])
    def add(i): torch.add(mids[i % ring], 1.0, out=dsts[i % ring])
    row = {"ring": ring, "footprint_MB": round(ring * (iw * ih + 2 * ow * oh) * 8 / 1e6)}
    for name, fn in (("easu", easu), ("rcas", rcas), ("pair", pair), ("copy", copy), ("add", add), ("rcas2", rcas), ("pair2", pair)):
        row[name + "_us"] = us(fn)
    print(json.dumps(row), flush=True)
    del srcs, mids, dsts
    torch.cuda.empty_cache()
