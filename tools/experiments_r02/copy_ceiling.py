#!/usr/bin/env python3
"""What a plain device-to-device copy of one 4K RGBA16F image (66.4 MB read + 66.4 MB written, RCAS's algorithmic traffic) takes on
this box, cold ring, next to RCAS itself: torch's elementwise copy kernel and hipMemcpyAsync D2D."""
import importlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
fsr = importlib.import_module("fidelityfx-fsr_amd"); fsr.load()
dev = torch.device("cuda", 0)
H, W = 2160, 3840
ring = 6
srcs = [torch.rand(H, W, 4, device=dev).half() for _ in range(ring)]
dsts = [torch.empty_like(s) for s in srcs]
timer = fsr.Timer()
rcon = fsr.FsrRcasCon(0.25)

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

row = {}
for rep in range(3):
    row.setdefault("torch_copy_us", []).append(us(lambda i: dsts[i % ring].copy_(srcs[i % ring])))
    row.setdefault("rcas_us", []).append(us(lambda i: fsr.rcas(srcs[i % ring], dsts[i % ring], con=rcon)))
    row.setdefault("torch_add_us", []).append(us(lambda i: torch.add(srcs[i % ring], 1.0, out=dsts[i % ring])))
    big_s = torch.stack(srcs[:4]); big_d = torch.empty_like(big_s)
    row.setdefault("torch_copy_4_frames_us_per_frame", []).append(round(us(lambda i: big_d.copy_(big_s), 100) / 4, 2))
    del big_s, big_d
mb = H * W * 8 * 2 / 1e6
row["bytes_MB"] = mb
row["TBps"] = {k: round(mb / min(v) / 1e6 * 1e6 / 1e6, 3) if isinstance(v, list) else None for k, v in row.items()}
print(json.dumps(row))
