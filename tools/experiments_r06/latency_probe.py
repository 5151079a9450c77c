"""Round 6 (VERDICT r5 Next 5): where do the microseconds of ONE 4K frame go — submit -> host-visible completion, GPU idle 1 ms before?

Variants, 1920x1080 -> 3840x2160 RGBA16F, default arithmetic and F-strict, two dispatches and the fused launch:
  sync        submit on a stream + hipStreamSynchronize                              (what bench.py's latency_us measures)
  query       submit + busy-poll hipEventQuery of an event recorded behind the frame (no interrupt / wake-up)
  flag        submit + hipStreamWriteValue32 to pinned host memory behind the frame, the host spins on the word
  graph       the frame's launches captured once into a hipGraph, hipGraphLaunch + each of the three completions above
Per variant: median host us (submit -> completion seen), median submit us (the call(s) returning), device us (events around the frame).
Also: the same with NO idle gap, and with the performance level forced to `high` when the sysfs file can be written (reported, never required).
"""
import ctypes
import importlib
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
fsr = importlib.import_module("fidelityfx-fsr_amd")
fsr.load()
hip = ctypes.CDLL("libamdhip64.so")
dev = torch.device("cuda", 0)
iw, ih, ow, oh = 1920, 1080, 3840, 2160
src = torch.from_numpy(fsr.frames.synthetic_frame(iw, ih, k=1)).to(dev)
mid = torch.empty(oh, ow, 4, dtype=torch.float16, device=dev)
dsts = [torch.empty(oh, ow, 4, dtype=torch.float16, device=dev) for _ in range(4)]
econ, rcon = fsr.FsrEasuCon(iw, ih, iw, ih, ow, oh), fsr.FsrRcasCon(0.25)
stream = torch.cuda.Stream()
flag_host = torch.zeros(16, dtype=torch.int32).pin_memory()
flag_ptr = ctypes.c_void_p(flag_host.data_ptr())
flag_np = flag_host.numpy()
hip.hipStreamWriteValue32.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint]
# the pinned word as the device sees it
dptr = ctypes.c_void_p()
hip.hipHostGetDevicePointer.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_void_p, ctypes.c_uint]
assert hip.hipHostGetDevicePointer(ctypes.byref(dptr), flag_ptr, 0) == 0


def frame(flags, fused, i):
    if fused:
        fsr.easu_rcas_fused(src, dsts[i % 4], easu_con=econ, rcas_con=rcon, flags=flags, stream=stream)
    else:
        fsr.easu(src, mid, con=econ, flags=flags, stream=stream)
        fsr.rcas(mid, dsts[i % 4], con=rcon, flags=flags, stream=stream)


def measure(submit, completion, n=200, gap=1e-3):
    """submit(i) enqueues the frame on `stream`; completion: 'sync' | 'query' | 'flag'"""
    ev0 = [torch.cuda.Event(enable_timing=True) for _ in range(n)]
    ev1 = [torch.cuda.Event(enable_timing=True) for _ in range(n)]
    host, sub = [], []
    for i in range(30):
        submit(i)
    stream.synchronize()
    seq = 0
    for i in range(n):
        if gap:
            time.sleep(gap)
        seq += 1
        t0 = time.perf_counter()
        ev0[i].record(stream)
        submit(i)
        ev1[i].record(stream)
        if completion == "flag":
            hip.hipStreamWriteValue32(ctypes.c_void_p(stream.cuda_stream), dptr, seq, 0)
        t1 = time.perf_counter()
        if completion == "sync":
            stream.synchronize()
        elif completion == "query":
            while not ev1[i].query():
                pass
        else:
            while flag_np[0] != seq:
                pass
        t2 = time.perf_counter()
        host.append(t2 - t0)
        sub.append(t1 - t0)
    stream.synchronize()
    devt = [a.elapsed_time(b) * 1e-3 for a, b in zip(ev0, ev1)]
    med = lambda v: float(np.median(v))
    return {"host_us": round(med(host) * 1e6, 1), "host_p90_us": round(float(np.percentile(host, 90)) * 1e6, 1), "submit_us": round(med(sub) * 1e6, 1),
            "device_us": round(med(devt) * 1e6, 1)}


def graph_of(flags, fused):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(stream):
        frame(flags, fused, 0)
        stream.synchronize()
        with torch.cuda.graph(g, stream=stream):
            frame(flags, fused, 0)
    return g


def run_all(tag):
    out = {}
    for math, flags in (("default", 0), ("strict", fsr.FLAG_MATH_STRICT)):
        for fused in (0, 1):
            key = "%s_%s" % (math, "fused" if fused else "two_pass")
            g = graph_of(flags, fused)

            def replay(i, g=g):
                with torch.cuda.stream(stream):
                    g.replay()
            res = {}
            for comp in ("sync", "query", "flag"):
                res["launch_" + comp] = measure(lambda i, flags=flags, fused=fused: frame(flags, fused, i), comp)
                res["graph_" + comp] = measure(replay, comp)
            res["launch_sync_back_to_back"] = measure(lambda i, flags=flags, fused=fused: frame(flags, fused, i), "sync", gap=0.0)
            res["graph_flag_back_to_back"] = measure(replay, "flag", gap=0.0)
            out[key] = res
            print(tag, key, json.dumps(res), flush=True)
    return out


if __name__ == "__main__":
    report = {"normal": run_all("normal")}
    # performance level `high` (report only): needs write access to the sysfs file
    lvl = None
    try:
        import glob
        for c in sorted(glob.glob("/sys/class/drm/card*/device/power_dpm_force_performance_level")):
            if os.path.exists(os.path.join(os.path.dirname(c), "gpu_metrics")):
                lvl = c
                break
        old = open(lvl).read().strip()
        open(lvl, "w").write("high")
        report["perf_level"] = {"file": lvl, "was": old, "set": open(lvl).read().strip()}
        report["perf_level_high"] = run_all("high")
        open(lvl, "w").write(old)
    except Exception as e:
        report["perf_level"] = {"file": lvl, "error": str(e)[:200]}
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "r06_latency_probe.json"), "w") as f:
        json.dump(report, f, indent=1)
