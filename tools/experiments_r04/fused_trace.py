#!/usr/bin/env python3
"""Timeline of the exact-2x fused launch on ONE 4K frame (experiment; needs variants/libfsr1_trace.so built from
tools/experiments_r04/fused_s2_trace.patch): every workgroup records wall_clock64() (100 MHz) at its start, after the staging of
its first step, after that step's EASU phase, after its RCAS phase, and at the end of its run, plus the CU it ran on.
Prints, per forced steps-per-run: launch duration from the trace, the phases' durations by dispatch order (first residency vs
the rest), how many workgroups are in flight over time, and when each CU runs dry (the tail)."""
import ctypes
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("FSR1_HIP_LIB", os.path.join(ROOT, "variants", "libfsr1_trace.so"))
import torch  # noqa: E402

fsr = importlib.import_module("fidelityfx-fsr_amd")
lib = fsr.load()
lib.fsr1_debug_trace_read.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
TICK_US = 0.01  # wall_clock64: 100 MHz


def run(in_w, in_h, frames, steps):
    out_w, out_h = 2 * in_w, 2 * in_h
    dev = torch.device("cuda", 0)
    ring = 8
    base = torch.from_numpy(fsr.frames.synthetic_frame(in_w, in_h, k=1)).to(dev)
    srcs = [torch.stack([torch.roll(base, shifts=(3 * s + f, 5 * s), dims=(0, 1)) for f in range(frames)]).contiguous() for s in range(ring)]
    dsts = [torch.empty(frames, out_h, out_w, 4, dtype=torch.float16, device=dev) for _ in range(ring)]
    econ, rcon = fsr.FsrEasuCon(in_w, in_h, in_w, in_h, out_w, out_h), fsr.FsrRcasCon(0.25)
    lib.fsr1_debug_fused_run_steps(steps)
    for i in range(400):  # clock ramp, steady power state
        fsr.easu_rcas_fused(srcs[i % ring], dsts[i % ring], easu_con=econ, rcas_con=rcon)
    torch.cuda.synchronize()
    tiles_x = -(-out_w // 62)
    tiles_y = -(-out_h // (16 * steps - 2))
    n = tiles_x * tiles_y * frames
    buf = np.zeros((65536, 8), np.uint64)
    reps = []
    for rep in range(5):
        for i in range(20):
            fsr.easu_rcas_fused(srcs[i % ring], dsts[i % ring], easu_con=econ, rcas_con=rcon)
        torch.cuda.synchronize()
        assert lib.fsr1_debug_trace_read(buf.ctypes.data, buf.nbytes) == 0
        reps.append(buf[:min(n, 65536)].astype(np.int64).copy())
    t = reps[-1]
    t0 = t[:, 0].min()
    start, staged, easu, rcas, end = [(t[:, k] - t0) * TICK_US for k in range(5)]
    dur = [float((r[:, 4].max() - r[:, 0].min()) * TICK_US) for r in reps]
    print("== %dx%d x%d -> %dx%d, %d step(s) per run: %d workgroups; launch (first start -> last end) %s us" %
          (in_w, in_h, frames, out_w, out_h, steps, n, " ".join("%.1f" % d for d in dur)))
    hw = t[:, 5]
    cu = ((hw >> 32) & 0xf) * 1000 + ((hw >> 13) & 0x7) * 100 + ((hw >> 12) & 1) * 50 + ((hw >> 8) & 0xf)  # xcc, se, sh, cu
    order = np.argsort(start)
    slots = 256 * 7
    for name, sel in (("first residency (dispatched first)", order[:slots]), ("second residency", order[slots:2 * slots]), ("the rest", order[2 * slots:])):
        if len(sel) == 0:
            continue
        q = lambda a: "%.2f / %.2f / %.2f" % tuple(np.percentile(a[sel], [10, 50, 90]))  # noqa: E731
        print("  %-36s n=%5d  start %s  stage %s  easu %s  rcas %s  step0 total %s  run %s  (us: p10 / p50 / p90)" %
              (name, len(sel), q(start), q(staged - start), q(easu - staged), q(rcas - easu), q(rcas - start), q(end - start)))
    tend = end.max()
    print("  workgroups in flight at t (us):", " ".join("%d:%d" % (x, int(((start <= x) & (end > x)).sum())) for x in np.arange(0, tend + 2, 2.0)))
    cus = np.unique(cu)
    last = np.array([end[cu == c].max() for c in cus])
    cnt = np.array([(cu == c).sum() for c in cus])
    print("  %d CUs seen; workgroups per CU min/median/max %d/%d/%d; CU runs dry at (us before the launch's end) p10/p50/p90/max: %s" %
          (len(cus), cnt.min(), np.median(cnt), cnt.max(), " / ".join("%.1f" % v for v in np.percentile(tend - last, [10, 50, 90, 100]))))
    busy = np.array([np.sum(end[cu == c] - start[cu == c]) for c in cus])
    print("  sum of workgroup lifetimes per CU / launch duration (average workgroups resident per CU): median %.2f, min %.2f" % (np.median(busy) / tend, busy.min() / tend))
    lib.fsr1_debug_fused_run_steps(0)


if __name__ == "__main__":
    for steps in (1, 2, 3, 5):
        run(1920, 1080, 1, steps)
    run(1920, 1080, 4, 1)
    run(1920, 1080, 4, 4)
