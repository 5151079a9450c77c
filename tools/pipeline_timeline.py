#!/usr/bin/env python3
"""Where the time of a pipelined frame stream goes, from a rocprofv3 --kernel-trace of bench.py (Next 7).

  python tools/pipeline_timeline.py DIR [--skip-first N] [--last N]

Reads *kernel_trace.csv (Start_Timestamp / End_Timestamp per dispatch, ns) and reports, over the steady part of the run:
  * per kernel: dispatches, mean / median duration (under overlap: a kernel's duration includes the time it shares the chip)
  * the wall time per frame (last end - first start over the frames in the window)
  * concurrency: the fraction of the wall time with 0 / 1 / 2 / 3+ kernels resident, and which kernels they are (E = EASU,
    R = RCAS, F = fused): "E+R" is an RCAS wave set running beside an EASU one
  * per queue: the gap between the end of a kernel and the start of the next kernel on the same queue (the kernel boundary)
  * the critical-resource view: sum over kernels of duration x (1 / number of kernels resident) = "chip time" per kernel class
"""
import csv
import glob
import statistics
import sys
from collections import defaultdict


def short(name):
    if "fused" in name:
        return "F"
    if "easu" in name:
        return "E"
    if "rcas" in name:
        return "R"
    return "?"


def main():
    d = sys.argv[1]
    skip = int(sys.argv[sys.argv.index("--skip-first") + 1]) if "--skip-first" in sys.argv else 400
    last = int(sys.argv[sys.argv.index("--last") + 1]) if "--last" in sys.argv else 1200
    rows = []
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if k == "?":
                continue
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), k, r.get("Queue_Id", "0"), r["Kernel_Name"]))
    rows.sort()
    rows = rows[skip:skip + last]
    if not rows:
        print("no kernels found")
        return
    t0, t1 = rows[0][0], max(r[1] for r in rows)
    wall = t1 - t0
    per = defaultdict(list)
    for s, e, k, q, n in rows:
        per[k].append(e - s)
    frames = max(len(per.get("E", [])), len(per.get("F", [])))
    print("window: %d dispatches, %d frames, %.1f us of wall time, %.2f us per frame" % (len(rows), frames, wall / 1e3, wall / 1e3 / max(frames, 1)))
    for k, v in sorted(per.items()):
        print("  kernel %s: %5d dispatches, duration mean %.2f us, median %.2f us, min %.2f, max %.2f" % (k, len(v), statistics.mean(v) / 1e3, statistics.median(v) / 1e3, min(v) / 1e3, max(v) / 1e3))
    # concurrency sweep
    ev = []
    for s, e, k, q, n in rows:
        ev.append((s, 1, k))
        ev.append((e, -1, k))
    ev.sort()
    active = defaultdict(int)
    hist = defaultdict(int)
    chip = defaultdict(float)
    prev = ev[0][0]
    for t, dlt, k in ev:
        if t > prev:
            n = sum(active.values())
            key = "+".join(sorted(kk for kk, c in active.items() for _ in range(c))) or "idle"
            hist[key] += t - prev
            if n:
                for kk, c in active.items():
                    chip[kk] += (t - prev) * c / n
            prev = t
        active[k] += dlt
    print("  residency (fraction of wall time):")
    for key, v in sorted(hist.items(), key=lambda kv: -kv[1]):
        print("    %-10s %6.3f  (%.2f us per frame)" % (key, v / wall, v / 1e3 / max(frames, 1)))
    print("  chip time per frame, splitting shared intervals evenly: " + ", ".join("%s %.2f us" % (k, v / 1e3 / max(frames, 1)) for k, v in sorted(chip.items())))
    # per-queue boundaries
    byq = defaultdict(list)
    for s, e, k, q, n in rows:
        byq[q].append((s, e, k))
    gaps = defaultdict(list)
    for q, v in byq.items():
        v.sort()
        for (s0, e0, k0), (s1, e1, k1) in zip(v, v[1:]):
            gaps[k0 + "->" + k1].append(s1 - e0)
    for key, v in sorted(gaps.items()):
        print("  same-queue boundary %s: median %.2f us, mean %.2f us (%d)" % (key, statistics.median(v) / 1e3, statistics.mean(v) / 1e3, len(v)))
    print("  queues: %d" % len(byq))


if __name__ == "__main__":
    main()
