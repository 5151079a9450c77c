#!/usr/bin/env python3
"""Does RCAS of frame i overlap with EASU of frame i+1 when the two passes are issued on two streams?

Serial: both dispatches on one stream (what fsr1_upscale does).  Overlapped: EASU on stream A, an event, RCAS on stream B
waiting for it; EASU of the next frame set starts while RCAS of the previous one runs.  Wall-clock per step over N steps.
"""
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402

fsr = importlib.import_module("fidelityfx-fsr_amd")
fsr.load()
dev = torch.device("cuda", 0)


def run(wl, n, math="f"):
    in_w, in_h, out_w, out_h, frames = bench.WORKLOADS[wl]
    flags = {"f": 0, "h": fsr.FLAG_MATH_PACKED_FP16}[math]
    set_bytes = (in_w * in_h + 2 * out_w * out_h) * 8 * frames
    ring = max(4, -(-320 * 2**20 // set_bytes))
    base = torch.from_numpy(fsr.frames.synthetic_frame(in_w, in_h, k=1)).to(dev)
    srcs = [torch.stack([torch.roll(base, shifts=(3 * s + f, 5 * s + 2 * f), dims=(0, 1)) for f in range(frames)]).contiguous() for s in range(ring)]
    mids = [torch.empty(frames, out_h, out_w, 4, dtype=torch.float16, device=dev) for _ in range(ring)]
    dsts = [torch.empty(frames, out_h, out_w, 4, dtype=torch.float16, device=dev) for _ in range(ring)]
    econ = fsr.FsrEasuCon(in_w, in_h, in_w, in_h, out_w, out_h)
    rcon = fsr.FsrRcasCon(0.25)

    def serial(i, _):
        s = i % ring
        fsr.easu(srcs[s], mids[s], con=econ, flags=flags)
        fsr.rcas(mids[s], dsts[s], con=rcon, flags=flags)

    def make_overlap(prio_b):
        sa = torch.cuda.Stream(device=dev)
        sb = torch.cuda.Stream(device=dev, priority=prio_b)
        ev_e = [torch.cuda.Event() for _ in range(ring)]
        ev_r = [torch.cuda.Event() for _ in range(ring)]
        state = {"first": True}

        def step(i, _):
            s = i % ring
            if i >= ring:
                sa.wait_event(ev_r[s])  # the intermediary slot is free once its previous RCAS has run
            fsr.easu(srcs[s], mids[s], con=econ, flags=flags, stream=sa)
            ev_e[s].record(sa)
            sb.wait_event(ev_e[s])
            fsr.rcas(mids[s], dsts[s], con=rcon, flags=flags, stream=sb)
            ev_r[s].record(sb)
        return step

    def half_batches(i, st):
        """one batch split into two halves on two streams (each half: EASU -> RCAS in stream order)"""
        s = i % ring
        h = frames // 2
        for k, stream in enumerate(st):
            sl = slice(k * h, (k + 1) * h if k == 0 else frames)
            fsr.easu(srcs[s][sl], mids[s][sl], con=econ, flags=flags, stream=stream)
            fsr.rcas(mids[s][sl], dsts[s][sl], con=rcon, flags=flags, stream=stream)

    def wall(fn, arg=None):
        t0 = time.perf_counter()
        i = 0
        while time.perf_counter() - t0 < 0.3:
            fn(i, arg); i += 1
            if i % 64 == 0:
                torch.cuda.synchronize()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n):
            fn(i, arg)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e6

    row = {"workload": wl, "math": math, "serial_us": round(wall(serial), 2)}
    row["overlap_us"] = round(wall(make_overlap(0)), 2)
    row["overlap_rcas_high_prio_us"] = round(wall(make_overlap(-1)), 2)
    if frames >= 2:
        st = (torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev))
        row["two_half_batches_us"] = round(wall(half_batches, st), 2)
    row["serial_again_us"] = round(wall(serial), 2)
    row["mpix_s_serial"] = round(out_w * out_h * frames / row["serial_us"], 1)
    row["mpix_s_best"] = round(out_w * out_h * frames / min(v for k, v in row.items() if k.endswith("_us")), 1)
    print(json.dumps(row), flush=True)


for wl, n in (("1080p_to_4k", 3000), ("1440p_to_4k", 3000), ("1440p_to_4k_x8", 400), ("4k_to_8k_x16", 60)):
    run(wl, n)
run("1080p_to_4k", 3000, "h")
