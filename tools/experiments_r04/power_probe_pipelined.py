#!/usr/bin/env python3
"""Package power and shader clock (rocm-smi sampled from a side thread) while the 1080p -> 4K pipelines loop for 6 s each: one in-order
stream against three, two dispatches and the fused launch.  Does filling the kernel boundaries cost clock?  (experiment)"""
import importlib
import json
import os
import re
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

fsr = importlib.import_module("fidelityfx-fsr_amd")
fsr.load()
dev = torch.device("cuda", 0)
in_w, in_h, out_w, out_h = 1920, 1080, 3840, 2160
ring = 16
base = torch.from_numpy(fsr.frames.synthetic_frame(in_w, in_h, k=1)).to(dev)
srcs = [torch.roll(base, shifts=(3 * s, 5 * s), dims=(0, 1)).contiguous().unsqueeze(0) for s in range(ring)]
dsts = [torch.empty(1, out_h, out_w, 4, dtype=torch.float16, device=dev) for _ in range(ring)]


def probe(streams, fused, flags, seconds=6.0):
    pipe = fsr.Pipeline(streams)
    samples, stop = [], threading.Event()

    def sample():
        while not stop.is_set():
            try:
                out = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True, timeout=10).stdout
                p = re.search(r"Power[^\n]*?:\s*([0-9.]+)", out)
                c = re.search(r"sclk clock level[^\n]*?\(([0-9.]+)Mhz\)", out)
                samples.append((float(p.group(1)) if p else None, float(c.group(1)) if c else None))
            except Exception:  # noqa: BLE001
                samples.append((None, None))
            time.sleep(0.3)
    th = threading.Thread(target=sample)
    th.start()
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < seconds:
        for _ in range(64):
            pipe.upscale(srcs[n % ring], dsts[n % ring], fused=fused, flags=flags)
            n += 1
        pipe.synchronize()
    dt = time.perf_counter() - t0
    stop.set()
    th.join()
    pipe.close()
    mid = samples[len(samples) // 4:] or samples
    pw = sorted(s[0] for s in mid if s[0] is not None)
    ck = sorted(s[1] for s in mid if s[1] is not None)
    print(json.dumps({"pipeline": "fused" if fused else "two dispatches", "math": "h" if flags else "f", "streams": streams, "us_per_frame_wall": round(dt / n * 1e6, 2),
                      "power_W_median": pw[len(pw) // 2] if pw else None, "sclk_MHz_median": ck[len(ck) // 2] if ck else None, "sclk_MHz_min": ck[0] if ck else None,
                      "samples": len(samples)}), flush=True)


for streams in (1, 3):
    probe(streams, 0, 0)
    probe(streams, 1, 0)
    probe(streams, 0, fsr.FLAG_MATH_PACKED_FP16)
