#!/usr/bin/env python3
"""Is RCAS faster when its input sits in the Infinity Cache, and does EASU's output land there?  (tuning probe)"""
import importlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
fsr = importlib.import_module("fidelityfx-fsr_amd"); fsr.load()
dev = torch.device("cuda", 0)
iw, ih, ow, oh = 1920, 1080, 3840, 2160
ring = 4
base = torch.from_numpy(fsr.frames.synthetic_frame(iw, ih, k=1)).to(dev)
srcs = [torch.roll(base, shifts=(3 * s, 5 * s), dims=(0, 1)).contiguous() for s in range(ring)]
mids = [torch.empty(oh, ow, 4, dtype=torch.float16, device=dev) for _ in range(ring)]
dsts = [torch.empty(oh, ow, 4, dtype=torch.float16, device=dev) for _ in range(ring)]
econ = fsr.FsrEasuCon(iw, ih, iw, ih, ow, oh); rcon = fsr.FsrRcasCon(0.25)
for s in range(ring): fsr.easu(srcs[s], mids[s], con=econ)
torch.cuda.synchronize()
t = fsr.Timer()
def ramp(fn):
    t0 = time.perf_counter(); i = 0
    while time.perf_counter() - t0 < 0.25:
        fn(i); i += 1
        if i % 64 == 0: torch.cuda.synchronize()
    torch.cuda.synchronize()
def loop(fn, n=300):
    ramp(fn); t.start()
    for i in range(n): fn(i)
    t.stop(); return t.elapsed_ms() / n * 1e3
def timed_second(first, second, n=200):
    """average time of `second` alone when it always runs right after `first` (events around `second` only)"""
    ramp(lambda i: (first(i), second(i)))
    tot = 0.0
    for i in range(n):
        first(i); t.start(); second(i); t.stop(); tot += t.elapsed_ms()
    return tot / n * 1e3
res = {}
res["rcas_same_buffers_hot"] = loop(lambda i: fsr.rcas(mids[0], dsts[0], con=rcon))
res["rcas_ring_cold"] = loop(lambda i: fsr.rcas(mids[i % ring], dsts[i % ring], con=rcon))
res["rcas_after_easu_wrote_mid"] = timed_second(lambda i: fsr.easu(srcs[i % ring], mids[i % ring], con=econ), lambda i: fsr.rcas(mids[i % ring], dsts[i % ring], con=rcon))
res["rcas_after_easu_streamed_mid"] = timed_second(lambda i: fsr.easu(srcs[i % ring], mids[i % ring], con=econ, flags=fsr.FLAG_OUTPUT_STREAMING), lambda i: fsr.rcas(mids[i % ring], dsts[i % ring], con=rcon))
scratch = torch.empty(oh, ow, 4, dtype=torch.float16, device=dev)
res["rcas_after_copy_read_mid"] = timed_second(lambda i: scratch.copy_(mids[i % ring]), lambda i: fsr.rcas(mids[i % ring], dsts[i % ring], con=rcon))
res["rcas_cached_output_hot"] = loop(lambda i: fsr.rcas(mids[0], dsts[0], con=rcon, flags=fsr.FLAG_OUTPUT_CACHED))
res["easu_ring"] = loop(lambda i: fsr.easu(srcs[i % ring], mids[i % ring], con=econ))
res["easu_same_buffers"] = loop(lambda i: fsr.easu(srcs[0], mids[0], con=econ))
res["easu_streaming_ring"] = loop(lambda i: fsr.easu(srcs[i % ring], mids[i % ring], con=econ, flags=fsr.FLAG_OUTPUT_STREAMING))
print(json.dumps({k: round(v, 2) for k, v in res.items()}))
