#!/usr/bin/env python3
"""Do independent frames on alternating HIP streams run faster than back to back on one?  (experiment)

The per-workgroup timeline (profiles/ab_r04/r4c1_fused_trace.log) puts a fused 4K launch at 59.6 us from its first workgroup's first
instruction to its last workgroup's last, while back-to-back launches on one stream take 65 us each: ~5 us per kernel boundary in
which the chip computes nothing (drain, cache write-back, the next dispatch's ramp).  Frames are independent, so consecutive
frames may overlap: frame i on stream i % S, each stream with its own intermediary for the two-dispatch pipeline."""
import importlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

fsr = importlib.import_module("fidelityfx-fsr_amd")
fsr.load()
dev = torch.device("cuda", 0)


def run(in_w, in_h, out_w, out_h, n=3000):
    ring = max(4, -(-(1 << 30) // ((in_w * in_h + out_w * out_h) * 8)))
    base = torch.from_numpy(fsr.frames.synthetic_frame(in_w, in_h, k=1)).to(dev)
    srcs = [torch.roll(base, shifts=(3 * s, 5 * s), dims=(0, 1)).contiguous().unsqueeze(0) for s in range(ring)]
    dsts = [torch.empty(1, out_h, out_w, 4, dtype=torch.float16, device=dev) for _ in range(ring)]
    econ, rcon = fsr.FsrEasuCon(in_w, in_h, in_w, in_h, out_w, out_h), fsr.FsrRcasCon(0.25)
    for n_streams in (1, 2, 3):
        streams = [torch.cuda.Stream() for _ in range(n_streams)]
        mids = [torch.empty(1, out_h, out_w, 4, dtype=torch.float16, device=dev) for _ in range(n_streams)]

        def fused(i):
            fsr.easu_rcas_fused(srcs[i % ring], dsts[i % ring], easu_con=econ, rcas_con=rcon, stream=streams[i % n_streams])

        def pair(i):
            st = streams[i % n_streams]
            fsr.easu(srcs[i % ring], mids[i % n_streams], con=econ, stream=st)
            fsr.rcas(mids[i % n_streams], dsts[i % ring], con=rcon, stream=st)

        for name, fn in (("fused", fused), ("two dispatches", pair)):
            res = []
            for rep in range(3):
                t0 = time.perf_counter()
                i = 0
                while time.perf_counter() - t0 < 0.25:
                    fn(i); i += 1
                    if i % 64 == 0:
                        torch.cuda.synchronize()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for i in range(n):
                    fn(i)
                torch.cuda.synchronize()
                res.append((time.perf_counter() - t0) / n * 1e6)
            print("%dx%d -> %dx%d  %-15s %d stream(s): %s us per frame" % (in_w, in_h, out_w, out_h, name, n_streams, " ".join("%.2f" % r for r in res)), flush=True)


if __name__ == "__main__":
    run(1920, 1080, 3840, 2160)
    run(2560, 1440, 3840, 2160)
    run(960, 540, 1920, 1080)
