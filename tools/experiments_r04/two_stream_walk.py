#!/usr/bin/env python3
"""With frames pipelined over two streams the tail of one fused launch overlaps the head of the next — does the column walk, which
loses on a single 4K frame because of its long tail (r4c1_fused_trace.log), pay off now?  (experiment)"""
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


def run(in_w, in_h, n=3000):
    out_w, out_h = 2 * in_w, 2 * in_h
    ring = max(4, -(-(1 << 30) // ((in_w * in_h + out_w * out_h) * 8)))
    base = torch.from_numpy(fsr.frames.synthetic_frame(in_w, in_h, k=1)).to(dev)
    srcs = [torch.roll(base, shifts=(3 * s, 5 * s), dims=(0, 1)).contiguous().unsqueeze(0) for s in range(ring)]
    dsts = [torch.empty(1, out_h, out_w, 4, dtype=torch.float16, device=dev) for _ in range(ring)]
    for rep in range(2):
        for n_streams in (2, 1):
            pipe = fsr.Pipeline(n_streams)
            for steps, tall in ((0, -1), (1, 0), (2, -1), (3, -1), (4, -1), (6, -1), (8, -1)):
                lib.fsr1_debug_fused_run_steps(steps)
                lib.fsr1_debug_fused_tall_tiles(tall)
                t0 = time.perf_counter()
                i = 0
                while time.perf_counter() - t0 < 0.2:
                    pipe.upscale(srcs[i % ring], dsts[i % ring], fused=1); i += 1
                    if i % 64 == 0:
                        pipe.synchronize()
                pipe.synchronize()
                t0 = time.perf_counter()
                for i in range(n):
                    pipe.upscale(srcs[i % ring], dsts[i % ring], fused=1)
                pipe.synchronize()
                us = (time.perf_counter() - t0) / n * 1e6
                print("%dx%d -> %dx%d fused, %d stream(s), steps %s: %.2f us per frame" % (in_w, in_h, out_w, out_h, n_streams, "rule (tall tile)" if steps == 0 else ("%d%s" % (steps, " (256-thread tile)" if steps == 1 else "")), us), flush=True)
            pipe.close()
    lib.fsr1_debug_fused_run_steps(0)
    lib.fsr1_debug_fused_tall_tiles(-1)


if __name__ == "__main__":
    run(1920, 1080)
    run(1280, 720)
