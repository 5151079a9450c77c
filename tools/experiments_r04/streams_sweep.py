#!/usr/bin/env python3
"""Pipelines of 1 / 2 / 3 / 4 streams over the workloads of the bench (experiment): is a third stream ever worse than two?
FSR1_HIP_LIB selects a variant library (e.g. the 64 x 32 EASU tile)."""
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
TAG = os.path.basename(os.environ.get("FSR1_HIP_LIB", "tree"))
if os.environ.get("FSR1_EASU_TALL"):
    lib.fsr1_debug_easu_tall_tiles(int(os.environ["FSR1_EASU_TALL"]))
QUICK = os.environ.get("SWEEP_QUICK") == "1"  # three streams only, the F two-dispatch workloads


def rate(pipe, fn, n):
    t0 = time.perf_counter()
    i = 0
    while time.perf_counter() - t0 < 0.2:
        fn(i); i += 1
        if i % 64 == 0:
            pipe.synchronize()
    pipe.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        fn(i)
    pipe.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


def run(in_w, in_h, out_w, out_h, frames, flags, fused, what, n):
    ring = max(4, -(-(1 << 30) // ((in_w * in_h + out_w * out_h) * 8 * frames)))
    base = torch.from_numpy(fsr.frames.synthetic_frame(in_w, in_h, k=1)).to(dev)
    srcs = [torch.stack([torch.roll(base, shifts=(3 * s + f, 5 * s), dims=(0, 1)) for f in range(frames)]).contiguous() for s in range(ring)]
    dsts = [torch.empty(frames, out_h, out_w, 4, dtype=torch.float16, device=dev) for _ in range(ring)]
    row = []
    for rep in range(2):
        for n_streams in ((3,) if QUICK else (1, 2, 3, 4)):
            pipe = fsr.Pipeline(n_streams)
            row.append("%d: %.2f" % (n_streams, rate(pipe, lambda i: pipe.upscale(srcs[i % ring], dsts[i % ring], fused=fused, flags=flags), n)))
            pipe.close()
    print("[%s] %dx%d -> %dx%d x%d %s, us per step by streams: %s" % (TAG, in_w, in_h, out_w, out_h, frames, what, "  ".join(row)), flush=True)
    del srcs, dsts
    torch.cuda.empty_cache()


if __name__ == "__main__":
    H = fsr.FLAG_MATH_PACKED_FP16
    run(1920, 1080, 3840, 2160, 1, 0, 0, "two dispatches", 2000)
    if QUICK:
        run(2560, 1440, 3840, 2160, 1, 0, 0, "two dispatches", 2000)
        run(2560, 1440, 3840, 2160, 8, 0, 0, "two dispatches", 200)
        run(3840, 2160, 7680, 4320, 4, 0, 0, "two dispatches", 100)
    elif TAG == "tree":
        run(1920, 1080, 3840, 2160, 1, 0, 1, "fused", 2000)
        run(1920, 1080, 3840, 2160, 1, H, 0, "two dispatches H", 1500)
        run(2560, 1440, 3840, 2160, 1, 0, 0, "two dispatches", 2000)
        run(960, 540, 1920, 1080, 1, 0, 0, "two dispatches", 4000)
        run(960, 540, 1920, 1080, 1, 0, 1, "fused", 4000)
        run(1280, 720, 2560, 1440, 1, 0, 2, "auto", 3000)
        run(2560, 1440, 3840, 2160, 8, 0, 0, "two dispatches", 200)
        run(3840, 2160, 7680, 4320, 4, 0, 1, "fused", 100)
    else:
        run(3840, 2160, 7680, 4320, 4, 0, 0, "two dispatches", 100)
