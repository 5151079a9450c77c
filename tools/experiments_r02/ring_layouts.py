#!/usr/bin/env python3
"""Pipeline time per frame for three buffer layouts: everything rotating over R sets (bench.py until now, R = 3), one reused
intermediary with inputs / outputs rotating over R sets (what an application with one intermediary texture does), for R = 3 / 12 / 24."""
import importlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
fsr = importlib.import_module("fidelityfx-fsr_amd"); fsr.load()
dev = torch.device("cuda", 0)
timer = fsr.Timer()

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

for (iw, ih, ow, oh) in ((1920, 1080, 3840, 2160), (2560, 1440, 3840, 2160)):
    econ = fsr.FsrEasuCon(iw, ih, iw, ih, ow, oh); rcon = fsr.FsrRcasCon(0.25)
    base = torch.from_numpy(fsr.frames.synthetic_frame(iw, ih, k=1)).to(dev)
    for ring in (3, 12, 24):
        srcs = [torch.roll(base, shifts=(3 * s, 5 * s), dims=(0, 1)).contiguous() for s in range(ring)]
        mids = [torch.empty(oh, ow, 4, dtype=torch.float16, device=dev) for _ in range(ring)]
        dsts = [torch.empty(oh, ow, 4, dtype=torch.float16, device=dev) for _ in range(ring)]
        row = {"shape": "%dx%d->%dx%d" % (iw, ih, ow, oh), "ring": ring}
        for name, mid_of in (("all_rotating", lambda i: mids[i % ring]), ("one_intermediary", lambda i: mids[0])):
            def easu(i): fsr.easu(srcs[i % ring], mid_of(i), con=econ)
            def rcas(i): fsr.rcas(mid_of(i), dsts[i % ring], con=rcon)
            def pair(i): easu(i); rcas(i)
            def fused(i): fsr.easu_rcas_fused(srcs[i % ring], dsts[i % ring], easu_con=econ, rcas_con=rcon)
            row[name] = {"easu_us": us(easu), "rcas_us": us(rcas), "pair_us": us(pair)}
        row["fused_us"] = us(fused)
        print(json.dumps(row), flush=True)
        del srcs, mids, dsts
        torch.cuda.empty_cache()
