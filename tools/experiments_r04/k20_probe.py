#!/usr/bin/env python3
"""The driver times K = 20 steps per region: how do 1 / 2 / 3 streams and the overlap-time geometry (64 x 32 EASU tiles, 16-row RCAS
strips, walking fused launch) do on regions that short, where the pipeline's fill and drain are a tenth of the region?  (experiment)
Mimics bench.py: a ramp, then R = 9 regions of K steps, each bracketed by a device synchronize; prints the median region."""
import importlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

fsr = importlib.import_module("fidelityfx-fsr_amd")
lib = fsr.load()
dev = torch.device("cuda", 0)
in_w, in_h, out_w, out_h = 1920, 1080, 3840, 2160
ring = max(4, -(-(1 << 30) // ((in_w * in_h + out_w * out_h) * 8)))
base = torch.from_numpy(fsr.frames.synthetic_frame(in_w, in_h, k=1)).to(dev)
srcs = [torch.roll(base, shifts=(3 * s, 5 * s), dims=(0, 1)).contiguous().unsqueeze(0) for s in range(ring)]
dsts = [torch.empty(1, out_h, out_w, 4, dtype=torch.float16, device=dev) for _ in range(ring)]


def med(xs):
    xs = sorted(xs)
    return xs[len(xs) // 2]


def regions(pipe, fused, k, flags=0, r=9):
    t0 = time.perf_counter()
    i = 0
    while time.perf_counter() - t0 < 0.2:
        pipe.upscale(srcs[i % ring], dsts[i % ring], fused=fused, flags=flags); i += 1
        if i % 64 == 0:
            pipe.synchronize()
    torch.cuda.synchronize()
    out = []
    for _ in range(r):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for j in range(k):
            pipe.upscale(srcs[(i + j) % ring], dsts[(i + j) % ring], fused=fused, flags=flags)
        torch.cuda.synchronize()
        out.append((time.perf_counter() - t0) / k * 1e6)
        i += k
    return med(out), min(out), max(out)


for rep in range(2):
    for fused, name in ((0, "two dispatches"), (1, "fused")):
        for streams in (1, 2, 3):
            for tall in ((-1, 0) if streams > 1 and fused == 0 else (-1,)):
                lib.fsr1_debug_easu_tall_tiles(tall)
                pipe = fsr.Pipeline(streams)
                row = []
                for k in (20, 200, 2000):
                    m, lo, hi = regions(pipe, fused, k)
                    row.append("K=%d: %.2f (%.2f-%.2f)" % (k, m, lo, hi))
                print("%s, %d stream(s)%s: us per step  %s" % (name, streams, ", 64x16 EASU tiles forced" if tall == 0 else "", "   ".join(row)), flush=True)
                pipe.close()
lib.fsr1_debug_easu_tall_tiles(-1)
