#!/usr/bin/env python3
"""Packed-fp16 pipelines under two-stream frame pipelining: two H dispatches vs the fused exact-2x H launch at several run lengths;
and the F two-dispatch pipeline with 8- vs 16-row RCAS strips (FSR1_HIP_LIB=variants/libfsr1_rcas16.so).  (experiment)"""
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


def rate(pipe, fn, n=2000):
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


def run(in_w, in_h, out_w, out_h, flags, what):
    ring = max(4, -(-(1 << 30) // ((in_w * in_h + out_w * out_h) * 8)))
    base = torch.from_numpy(fsr.frames.synthetic_frame(in_w, in_h, k=1)).to(dev)
    srcs = [torch.roll(base, shifts=(3 * s, 5 * s), dims=(0, 1)).contiguous().unsqueeze(0) for s in range(ring)]
    dsts = [torch.empty(1, out_h, out_w, 4, dtype=torch.float16, device=dev) for _ in range(ring)]
    for rep in range(2):
        for n_streams in (2, 1):
            pipe = fsr.Pipeline(n_streams)
            row = ["%dx%d -> %dx%d %s, %d stream(s):" % (in_w, in_h, out_w, out_h, what, n_streams)]
            row.append("two dispatches %.2f" % rate(pipe, lambda i: pipe.upscale(srcs[i % ring], dsts[i % ring], fused=0, flags=flags)))
            if out_w == 2 * in_w:
                for steps in (1, 2, 4, 6):
                    lib.fsr1_debug_fused_run_steps(steps)
                    row.append("fused S=%d %.2f" % (steps, rate(pipe, lambda i: pipe.upscale(srcs[i % ring], dsts[i % ring], fused=1, flags=flags))))
                lib.fsr1_debug_fused_run_steps(0)
            else:
                row.append("fused %.2f" % rate(pipe, lambda i: pipe.upscale(srcs[i % ring], dsts[i % ring], fused=1, flags=flags)))
            print(" ".join(row), "us per frame", flush=True)
            pipe.close()


if __name__ == "__main__":
    tag = os.path.basename(os.environ.get("FSR1_HIP_LIB", "tree"))
    if len(sys.argv) > 1 and sys.argv[1] == "f":
        run(1920, 1080, 3840, 2160, 0, "F [%s]" % tag)
        run(2560, 1440, 3840, 2160, 0, "F [%s]" % tag)
    else:
        run(1920, 1080, 3840, 2160, fsr.FLAG_MATH_PACKED_FP16, "H")
        run(2560, 1440, 3840, 2160, fsr.FLAG_MATH_PACKED_FP16, "H")
