#!/usr/bin/env python3
"""Headline benchmark: upscaled megapixels/s of the FSR 1.0 hot path (EASU + RCAS) on MI355X.

  python bench.py --gpus N --steps K --warmup W
  N > 1, either launch style gives the same run and ONE JSON line:
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
    python bench.py --gpus N ...      (no WORLD_SIZE in the environment: bench.py starts the N ranks itself, the same way)

A "step" is one pass of the hot path over one batch of synthetic frames that are already resident
in HBM.  Default workload = BASELINE.json configs[1]: 1920x1080 -> 3840x2160, EASU + RCAS, RGBA16F
storage, one frame per step and per GPU, "F-strict" arithmetic (round 6, --math strict) = fp32 math whose
EASU image is bit-identical to the reference's CPU-evaluated FsrEasuF and whose final image is within
1 binary16 ULP of the reference chain FsrEasuF -> RTNE -> FsrRcasF: north_star's tolerance end to end
(--math f is the faster default arithmetic rounds 1-5 quoted: per stage within 1 ULP, end to end
99.99 % within 1 ULP; the line carries it as also_measured.default_arithmetic_*).  Steps rotate over a ring of distinct inputs and
outputs of more than 1 GiB (four times the 256 MiB Infinity Cache), so every step reads its frame from
HBM and writes its result to HBM like a video pipeline would; the EASU->RCAS intermediary is one
buffer reused by every step, as the sample's single intermediary texture is.
Frames are independent, so N GPUs run N independent streams (weak scaling); the only collective is
the reduction of the throughput counters.  Rank 0 prints ONE JSON line.
"""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "oracle")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6.3 TB/s is the measured copy ceiling
# VALU issue peak: 256 CUs x 4 SIMDs, one wave64 fp32 instruction per 2 cycles at 2.4 GHz (MI355X_MICROARCH.md:
# "v_fma_f32 (wave64) 2 cyc (SIMD-32)") = 1228.8 G wave-instructions/s = 157.3 TFLOP/s of fp32 FMA
VALU_PEAK_GINST = 256 * 4 * 2.4 / 2.0
# BASELINE.md §1 (docs/FidelityFX-FSR-Overview-Integration.pdf p.9): "FSR performance overhead" — EASU + RCAS, FP16 path —
# at a 4K target is <= 0.40 ms on the fastest tier the reference measured (RX 6800 XT / RTX 3080), for all four presets,
# i.e. >= 20 700 upscaled Mpix/s.  The only published figure for this metric: an upper bound on time, on other hardware.
PUBLISHED_4K_MPIX_S = 3840 * 2160 / 0.40e-3 / 1e6
# same slide, 1440p target: <= 0.20 ms on the same tier = >= 18 432 Mpix/s
PUBLISHED_1440P_MPIX_S = 2560 * 1440 / 0.20e-3 / 1e6

WORKLOADS = {
    # name: (in_w, in_h, out_w, out_h, frames per step per GPU)
    "1080p_to_4k": (1920, 1080, 3840, 2160, 1),          # BASELINE configs[1] (and [3] when --pipeline fused)
    "540p_to_1080p": (960, 540, 1920, 1080, 1),          # configs[0] shape
    "270p_to_540p": (480, 270, 960, 540, 1),             # a launch-bound size (see --graph)
    "720p_to_1440p": (1280, 720, 2560, 1440, 1),         # the reference's other published target (PDF p.9: 1440p output)
    "720p_to_1080p": (1280, 720, 1920, 1080, 1),         # 1.5x at a launch-bound size (`auto` takes the fused launch below 3 Mpixel)
    "831p_to_1080p": (1477, 831, 1920, 1080, 1),         # 1.3x at the same size
    "1440p_to_4k": (2560, 1440, 3840, 2160, 1),          # 1.5x "Quality", one frame
    "1270p_to_4k": (2259, 1270, 3840, 2160, 1),          # 1.7x "Balanced" (PDF p.10 true-ratio shape)
    "1662p_to_4k": (2954, 1662, 3840, 2160, 1),          # 1.3x "Ultra Quality"
    "1440p_to_4k_x8": (2560, 1440, 3840, 2160, 8),       # configs[2]: 64 frames over 8 GPUs
    "4k_to_8k_x16": (3840, 2160, 7680, 4320, 16),        # configs[4]: 128 frames over 8 GPUs
    "4k_to_8k": (3840, 2160, 7680, 4320, 1),             # one 8K frame
    "4k_to_8k_x4": (3840, 2160, 7680, 4320, 4),          # (sizes between one 8K frame and configs[4]'s shard: where `auto` switches for packed fp16)
    "4k_to_8k_x8": (3840, 2160, 7680, 4320, 8),
    "1080p_to_4k_x2": (1920, 1080, 3840, 2160, 2),       # two / four / eight 4K frames per launch (the exact-2x shape as small batches)
    "1080p_to_4k_x4": (1920, 1080, 3840, 2160, 4),
    "1080p_to_4k_x8": (1920, 1080, 3840, 2160, 8),
}


def reduce_counters(frames, pixels, seconds, device):
    """Whole-job counters: SUM of frames/pixels, MAX of seconds over ranks (identity without a process group).
    This is the only collective of the multi-GPU path (RCCL over xGMI on GPUs, gloo in the CPU tests)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return {"frames": int(frames), "pixels": int(pixels), "seconds": float(seconds)}
    s = torch.tensor([float(frames), float(pixels)], dtype=torch.float64, device=device)
    m = torch.tensor([float(seconds)], dtype=torch.float64, device=device)
    dist.all_reduce(s, op=dist.ReduceOp.SUM)
    dist.all_reduce(m, op=dist.ReduceOp.MAX)
    return {"frames": int(round(s[0].item())), "pixels": int(round(s[1].item())), "seconds": float(m[0].item())}


def gather_seconds(seconds, device):
    """Every rank's own time for its K steps, in rank order (one more counters-only collective)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return [float(seconds)]
    mine = torch.tensor([float(seconds)], dtype=torch.float64, device=device)
    out = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(out, mine)
    return [float(t.item()) for t in out]


def median(xs):
    xs = sorted(float(x) for x in xs)
    n = len(xs)
    return xs[n // 2] if n % 2 else 0.5 * (xs[n // 2 - 1] + xs[n // 2])


class Telemetry:
    """Shader clock and package power of ONE GPU, sampled by a host thread from the amdgpu driver's sysfs files while the timed
    regions run (Next 3a: the F kernels run at the package power cap, so an 8-GPU curve is decided by the chassis as
    much as by the code — the line has to say what clock and power every rank saw).  Files, under the device's PCI node
    (/sys/bus/pci/devices/<domain:bus:dev.fn>/hwmon/hwmon*/): freq1_input (sclk, Hz), power1_average or power1_input (uW).
    Everything is best effort: a file that is missing or unreadable leaves that figure null; the run is never failed by it."""

    def __init__(self, torch, dev_index, period=0.002):
        import glob
        import threading
        self.period = period
        self.freq = self.power = None
        self.source = None
        self.samples = []     # (t, mhz or None, watts or None)
        self.spans = {}       # label -> [(t0, t1), ...]
        self._open = {}
        self._stop = threading.Event()
        self._thread = None
        node = None
        try:
            try:
                p = torch.cuda.get_device_properties(dev_index)
                bdf = "%04x:%02x:%02x.0" % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id)
                if os.path.isdir("/sys/bus/pci/devices/" + bdf):
                    node, self.source = "/sys/bus/pci/devices/" + bdf, "sysfs hwmon of " + bdf
            except Exception:
                pass
            if node is None:  # no PCI ids from the runtime: the dev_index-th AMD GPU with a hwmon node, in PCI order
                def amd_gpu(c):  # (partition nodes — /sys/devices/platform/amdgpu_xcp_* — have no vendor file: skipped)
                    try:
                        return open(os.path.join(c, "vendor")).read().strip() == "0x1002" and bool(glob.glob(os.path.join(c, "hwmon", "hwmon*")))
                    except OSError:
                        return False
                cands = sorted(set(os.path.realpath(c) for c in glob.glob("/sys/class/drm/card*/device") if amd_gpu(c)))
                if dev_index < len(cands):
                    node, self.source = cands[dev_index], "sysfs hwmon of %s (device %d in PCI order)" % (os.path.basename(cands[dev_index]), dev_index)
            if node:
                for hw in sorted(glob.glob(os.path.join(node, "hwmon", "hwmon*"))):
                    f = os.path.join(hw, "freq1_input")
                    if self.freq is None and os.path.exists(f):
                        self.freq = f
                    for name in ("power1_average", "power1_input"):
                        f = os.path.join(hw, name)
                        if self.power is None and os.path.exists(f) and self._read(f) is not None:
                            self.power = f
        except Exception:
            pass
        # static facts of the device the figures are read against: the package power cap, the compute-partition mode (the kernels' XCD
        # swizzle assumes SPX: 8 XCDs behind one device) and the NUMA node / local CPUs of its PCIe root
        self.node = node
        self.power_cap_w = self.partition = self.numa_node = self.local_cpus = None
        try:
            if self.power:
                v = self._read(os.path.join(os.path.dirname(self.power), "power1_cap"))
                self.power_cap_w = round(v / 1e6, 1) if v else None
            if self.node:
                for attr, name in (("partition", "current_compute_partition"), ("numa_node", "numa_node"), ("local_cpus", "local_cpulist")):
                    try:
                        setattr(self, attr, open(os.path.join(self.node, name)).read().strip())
                    except OSError:
                        pass
        except Exception:
            pass
        if self.freq or self.power:
            self._thread = threading.Thread(target=self._run, daemon=True)
            self._thread.start()

    @staticmethod
    def _read(path):
        try:
            with open(path) as f:
                return float(f.read().split()[0])
        except Exception:
            return None

    def _run(self):
        while not self._stop.is_set():
            hz = self._read(self.freq) if self.freq else None
            uw = self._read(self.power) if self.power else None
            self.samples.append((time.perf_counter(), hz / 1e6 if hz else None, uw / 1e6 if uw else None))
            self._stop.wait(self.period)

    def begin(self, label):
        self._open[label] = time.perf_counter()

    def end(self, label):
        t0 = self._open.pop(label, None)
        if t0 is not None:
            self.spans.setdefault(label, []).append((t0, time.perf_counter()))

    def close(self):
        self._stop.set()
        if self._thread is not None:
            self._thread.join(timeout=1.0)

    def violations(self, dev_index):
        """The SMU's throttle ("violation") accumulators of this GPU from `amd-smi metric --violation --json` (best effort, ~1 s of host time,
        called outside every timed region): a dict of the numeric fields, or None.  The DIFFERENCE of two calls around a load window says
        which limiter was active during it (ppt = the package power cap, thermal, ...)."""
        import shutil
        import subprocess
        exe = shutil.which("amd-smi") or ("/opt/rocm/bin/amd-smi" if os.path.exists("/opt/rocm/bin/amd-smi") else None)
        if not exe:
            return None
        try:
            out = subprocess.run([exe, "metric", "-g", str(dev_index), "--violation", "--json"], capture_output=True, text=True, timeout=20).stdout
            doc = json.loads(out[out.index("{") if "{" in out and (out.lstrip()[:1] != "[") else out.index("["):])
            flat = {}

            def walk(prefix, v):
                if isinstance(v, dict):
                    if set(v) >= {"value"} and isinstance(v.get("value"), (int, float)):
                        flat[prefix] = v["value"]
                    else:
                        for k, x in v.items():
                            walk((prefix + "." if prefix else "") + str(k), x)
                elif isinstance(v, list):
                    for i, x in enumerate(v):
                        walk(prefix if len(v) == 1 else "%s[%d]" % (prefix, i), x)
                elif isinstance(v, (int, float)) and not isinstance(v, bool):
                    flat[prefix] = v
            walk("", doc)
            flat = {k.split("violation_status.")[-1]: v for k, v in flat.items() if "gpu" != k}
            return flat or None
        except Exception:
            return None

    def summary(self, label):
        """(median MHz, mean W, samples) over the spans recorded under `label`; None where nothing was read."""
        spans = self.spans.get(label, [])
        sel = [s for s in self.samples if any(t0 <= s[0] <= t1 for t0, t1 in spans)]
        mhz = [s[1] for s in sel if s[1]]
        w = [s[2] for s in sel if s[2]]
        return (round(median(mhz), 0) if mhz else None, round(sum(w) / len(w), 0) if w else None, len(sel))


def gather_pairs(a, b, device):
    """Every rank's (a, b) in rank order — per-rank clock and power; NaN stands for "not read" (one counters-only collective)."""
    import torch
    import torch.distributed as dist
    nan = float("nan")
    mine = torch.tensor([nan if a is None else float(a), nan if b is None else float(b)], dtype=torch.float64, device=device)
    if not (dist.is_available() and dist.is_initialized()):
        rows = [mine.tolist()]
    else:
        out = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
        dist.all_gather(out, mine)
        rows = [t.tolist() for t in out]
    clean = lambda v: None if v != v else v  # noqa: E731
    return [clean(r[0]) for r in rows], [clean(r[1]) for r in rows]


def gather_row(values, device):
    """Every rank's row of floats (None = "not read") in rank order: one counters-only collective."""
    import torch
    import torch.distributed as dist
    nan = float("nan")
    mine = torch.tensor([nan if v is None else float(v) for v in values], dtype=torch.float64, device=device)
    if not (dist.is_available() and dist.is_initialized()):
        rows = [mine.tolist()]
    else:
        out = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
        dist.all_gather(out, mine)
        rows = [t.tolist() for t in out]
    return [[None if v != v else v for v in r] for r in rows]


TELEMETRY_WINDOW_S = 1.2
TELEMETRY_MIN_SAMPLES = 200  # a clock / power figure from fewer samples of the load is marked unreliable (round 5's K = 20 line: 57 samples, 20 % low)


def telemetry_block(rows, source, period_ms, window_s, partition, extra=None):
    """The line's `telemetry` object from every rank's (mhz, watts, samples, cap_w, ppt_delta, thermal_delta) row."""
    samples = [int(r[2]) if r[2] is not None else 0 for r in rows]
    caps = [r[3] for r in rows]
    t = {"source": source, "period_ms": period_ms,
         "window": "a dedicated window of %.1f s of the headline's pipelined load per rank, after the clock ramp and before the warm-up and the timed regions "
                   "(outside all of them): long enough for >= %d samples whatever --steps is" % (window_s, TELEMETRY_MIN_SAMPLES),
         "samples_per_rank": samples, "samples_rank0": samples[0] if samples else 0,
         "reliable": bool(samples) and min(samples) >= TELEMETRY_MIN_SAMPLES,
         "power_cap_w": caps,
         "at_power_cap": [None if (r[1] is None or r[3] is None) else bool(r[1] >= 0.95 * r[3]) for r in rows],
         "throttle_accumulator_delta": {"ppt": [r[4] for r in rows], "thermal": [r[5] for r in rows],
                                        "note": "SMU violation accumulators (amd-smi metric --violation) after minus before the window, per rank: which limiter was "
                                                "active under the load; null where amd-smi does not report them"},
         "compute_partition": partition}
    if not t["reliable"]:
        t["unreliable"] = "fewer than %d samples of the load on at least one rank: per_rank_mhz / per_rank_watts under-read a short load" % TELEMETRY_MIN_SAMPLES
    if extra:
        t.update(extra)
    return t


def parse_cpulist(text):
    """"0-63,128-191" -> the set of CPU numbers (sysfs local_cpulist syntax)"""
    cpus = set()
    for part in (text or "").strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus.update(range(int(lo), int(hi or lo) + 1))
    return cpus


def bind_to_gpu_cpus(tele, enabled=True):
    """Pins this rank's process to the CPUs local to its GPU's PCIe root (sysfs local_cpulist / numa_node of the device: VERDICT r5, weak 8
    — every rank submits ~16 k steps/s through ctypes, and on a two-socket host half the ranks would otherwise launch across the socket
    link).  Best effort; returns what the line prints as config.cpu_binding."""
    info = {"numa_node": tele.numa_node, "local_cpus": tele.local_cpus, "applied": False}
    try:
        info["_before"] = os.sched_getaffinity(0)  # (restored around the cpu_baseline leg, which is meant to use every host core; not printed)
        cpus = parse_cpulist(tele.local_cpus) & os.sched_getaffinity(0)
        if enabled and cpus:
            os.sched_setaffinity(0, cpus)
            info["applied"] = True
            info["cpus_in_mask"] = len(cpus)
    except (OSError, ValueError, AttributeError) as e:
        info["error"] = str(e)[:80]
    return info


def submission_ceiling(fsr, torch, device, streams, seconds=0.25):
    """Steps per second ONE rank's host thread can push through fsr1_pipeline_upscale (ctypes, the bench's own path): 16 x 16 -> 32 x 32
    frames, whose kernels take no time to speak of, submitted for `seconds` (synchronising every 256 so the queues stay bounded), two
    dispatches per step like the headline.  Compared with what the headline needs (1 / ms_per_step): the host must not be what a rank —
    or eight of them on one host — waits for (Next 4)."""
    src = torch.zeros(1, 16, 16, 4, dtype=torch.float16, device=device)
    dsts = [torch.zeros(1, 32, 32, 4, dtype=torch.float16, device=device) for _ in range(max(2, streams))]
    pipe = fsr.Pipeline(max(1, streams), managed=False)
    pipe.reserve(32 * 32 * 8)
    for i in range(64):
        pipe.upscale(src, dsts[i % len(dsts)], sharpness=0.25, use_rcas=True, fused=0)
    pipe.synchronize()
    n, t0 = 0, time.perf_counter()
    busy = 0.0
    while True:
        t1 = time.perf_counter()
        for i in range(256):
            pipe.upscale(src, dsts[i % len(dsts)], sharpness=0.25, use_rcas=True, fused=0)
        busy += time.perf_counter() - t1
        pipe.synchronize()
        n += 256
        if time.perf_counter() - t0 >= seconds:
            break
    pipe.close()
    return {"steps_per_s": round(n / busy, 0), "us_per_step": round(busy / n * 1e6, 2), "steps": n,
            "what": "host time of fsr1_pipeline_upscale (two dispatches, %d streams) per step through ctypes, 16x16 -> 32x32 frames" % max(1, streams)}


def reduce_regions(region_seconds, device):
    """Elementwise MAX over ranks of the R region times (one counters-only collective): region r of the job took as long as its
    slowest rank's region r."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return [float(t) for t in region_seconds]
    m = torch.tensor([float(t) for t in region_seconds], dtype=torch.float64, device=device)
    dist.all_reduce(m, op=dist.ReduceOp.MAX)
    return [float(t) for t in m.tolist()]


STOPWATCH_SLACK = 1.03


def check_stopwatch(line):
    """Self-consistency of a bench line (Weak 1): kernels that run back to back on one stream cannot take longer
    than the step that contains them.  Checked: sum of kernels[*].avg_kernel_us <= 1.03 x ms_per_step for the two dispatches,
    and also_measured.fused.avg_kernel_us <= 1.03 x its own ms_per_step — "the step" being the one on ONE in-order stream
    (config.one_stream / also_measured.fused.one_stream when the line was taken with --streams > 1: pipelined steps overlap and are
    shorter than the kernels they contain).  On violation the line is marked
    "stopwatch_suspect": true, says which inequality failed, and every figure derived from the suspect kernel times (`frac`,
    `achieved`, `hbm_frac`, `valu_frac`, `read_only`, `valu`, `cold_input.frac`) is removed rather than printed wrong.
    Returns the list of violations (empty = consistent)."""
    bad = []
    step_us = ((line.get("config") or {}).get("one_stream") or {}).get("ms_per_step", line.get("ms_per_step", 0.0)) * 1e3  # the in-order step
    kern = line.get("kernels") or {}
    total = sum(k.get("avg_kernel_us", 0.0) for k in kern.values())
    if kern and step_us > 0 and total > STOPWATCH_SLACK * step_us:
        bad.append("sum of kernels[*].avg_kernel_us = %.2f us > %.2f x ms_per_step = %.2f us" % (total, STOPWATCH_SLACK, step_us))
    fused = (line.get("also_measured") or {}).get("fused") or {}
    fused_step_ms = (fused.get("one_stream") or {}).get("ms_per_step", fused.get("ms_per_step", 0))
    if "avg_kernel_us" in fused and fused_step_ms > 0 and fused["avg_kernel_us"] > STOPWATCH_SLACK * fused_step_ms * 1e3:
        bad.append("also_measured.fused.avg_kernel_us = %.2f us > %.2f x its (one-stream) ms_per_step = %.2f us"
                   % (fused["avg_kernel_us"], STOPWATCH_SLACK, fused_step_ms * 1e3))
    if bad:
        line["stopwatch_suspect"] = True
        line["stopwatch_violations"] = bad

        def strip(r):
            for k in ("frac", "achieved", "read_only", "valu", "hbm_frac", "valu_frac"):
                r.pop(k, None)
            if isinstance(r.get("cold_input"), dict):
                r["cold_input"].pop("frac", None)
        if isinstance(line.get("roofline"), dict):
            strip(line["roofline"])
        for r in kern.values():
            strip(r)
        if fused:
            strip(fused)
    else:
        line["stopwatch_suspect"] = False
    return bad


def free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(argv, n):
    """`python bench.py --gpus N` with no WORLD_SIZE in the environment: start the N ranks the way the driver's other launch
    style does — python -m torch.distributed.run --nnodes=1 --nproc-per-node N on 127.0.0.1 — and hand back its exit code.
    Rank 0's JSON line reaches this process's stdout unchanged (torchrun passes the workers' stdout through)."""
    import subprocess
    port = free_port()
    env = dict(os.environ, FSR1_BENCH_SELF_LAUNCHED="1", MASTER_ADDR="127.0.0.1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL across processes needs it on these hosts
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    return subprocess.call(cmd, env=env)


def stub_main(args, world, rank):
    """--stub: the multi-rank plumbing of this file with no GPU and no kernel — rendezvous, barriers, K "steps" of 1 ms of
    sleep, the counters-only collectives, rank 0's JSON line — over gloo on CPU.  A test fixture (tests/test_shard.py): the
    line says data = "stub" and carries no roofline."""
    import torch
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")
    device = torch.device("cpu")
    in_w, in_h, out_w, out_h, frames = WORKLOADS[args.workload]
    own = []
    for _ in range(args.regions):  # the same R-region bracket as the real run: barrier, K steps, clock, barrier
        if world > 1:
            dist.barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            time.sleep(1e-3)
        own.append(time.perf_counter() - t0)
        if world > 1:
            dist.barrier()
    regions = reduce_regions(own, device)
    seconds = median(regions)
    total = reduce_counters(frames * args.steps, frames * args.steps * out_w * out_h, median(own), device)
    total["seconds"] = seconds
    per_rank = gather_seconds(median(own), device)
    if rank == 0:
        print(json.dumps({"metric": "stub (no GPU work): plumbing of bench.py --gpus N", "value": round(total["pixels"] / total["seconds"] / 1e6, 1),
                          "unit": "Mpix/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": round(total["seconds"] * 1e3 / args.steps, 5), "higher_is_better": True, "scaling": "weak",
                          "vs_baseline": None, "dtype": "none", "data": "stub",
                          "config": {"workload": args.workload, "launch": launch_style(), "collective_backend": "gloo" if world > 1 else None,
                                     "world_size_seen": dist.get_world_size() if world > 1 else 1, "frames_total": total["frames"],
                                     "regions": args.regions,
                                     "region_ms_per_step": {"median": round(seconds * 1e3 / args.steps, 5), "min": round(min(regions) * 1e3 / args.steps, 5),
                                                            "max": round(max(regions) * 1e3 / args.steps, 5)}},
                          "per_rank_seconds": [round(t, 6) for t in per_rank]}), flush=True)
    if world > 1:
        dist.destroy_process_group()


def launch_style():
    if os.environ.get("FSR1_BENCH_SELF_LAUNCHED") == "1":
        return "self-launched: python bench.py --gpus N started torch.distributed.run itself"
    return "torch.distributed.run" if "WORLD_SIZE" in os.environ else "single process"


def pmc_traffic(fsr, workload, pipeline, kernel_name, math="f", storage="rgba16f"):
    """HBM bytes per launch of `kernel_name` from the committed rocprofv3 PMC summary of this very configuration
    (profiles/*.json, written by tools/gpu_profile.sh -> tools/prof_summary.py from separate --pmc passes of this
    command; FETCH_SIZE doubled per MI355X_MICROARCH.md's gfx950 correction, WRITE_SIZE as reported).  PMC counters
    cannot be collected from inside the timed run.  A summary is quoted only if it was taken of the kernel sources
    that are running now (`source_hash` over csrc/ + include/): otherwise `traffic` is null and the stale file is named.
    Returns (traffic_bytes or None, source path or None, SQ_INSTS_VALU or None, stale path or None)."""
    import glob
    now = fsr._lib.source_hash()
    best, stale = None, None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*.json"))):
        try:
            d = json.load(open(f))
        except (OSError, ValueError):
            continue
        if d.get("workload") != workload or d.get("pipeline") != pipeline or d.get("math", "f") != math or d.get("storage", "rgba16f") != storage:
            continue
        for k, v in d.get("kernels", {}).items():
            names = ["::%s%s_kernel<" % (kernel_name, "_h" if math == "h" else "")] + (["::fused_s2_kernel<"] if kernel_name == "fused" else [])
            if any(nm in k + "<" for nm in names) and "traffic_bytes" in v.get("hbm_per_launch", {}):
                if d.get("source_hash") != now:
                    stale = os.path.relpath(f, ROOT)
                    continue
                # several variants of one kernel template may be in a trace (EXACT, generic ...): the pipeline's own is the one launched most
                cand = (v.get("calls", 0), v["hbm_per_launch"]["traffic_bytes"], os.path.relpath(f, ROOT), v.get("pmc_avg_per_launch", {}).get("SQ_INSTS_VALU"))
                if best is None or cand[0] > best[0]:
                    best = cand
    if best:
        return best[1:] + (None,)
    return (None, None, None, stale)


def cpu_baseline(fsr, in_w, in_h, out_w, out_h, target_seconds=12.0, keep=None):
    """The reference path on the host cores: oracle/_ref (the reference headers compiled verbatim; kind
    "reference") when it travelled with the tree, else the plain-C restatement (kind "port"); OpenMP over
    output rows on all host cores; whole frames of the same workload (EASU-F then RCAS-F), repeated until about
    `target_seconds` of wall time have been spent (a bounded sample: one frame when a frame takes that long).
    keep: a dict that receives the first frame (binary16 input, the reference chain's intermediary and final image) for the line's `parity` block."""
    import numpy as np
    import cpu_oracle
    o = cpu_oracle.ref() if cpu_oracle.have_ref() else cpu_oracle.port()
    con = o.FsrEasuCon(in_w, in_h, in_w, in_h, out_w, out_h)
    rc = o.FsrRcasCon(0.25)
    t_easu = t_rcas = 0.0
    frames = 0
    t_begin = time.perf_counter()
    while True:
        img = fsr.frames.synthetic_frame(in_w, in_h, k=frames, dtype=np.float32)
        t0 = time.perf_counter()
        mid = o.easu_f(img, out_w, out_h, con, 0)
        t1 = time.perf_counter()
        mid = mid.astype(np.float16).astype(np.float32)  # the two-pass intermediary is RGBA16F
        t2 = time.perf_counter()
        out = o.rcas_f(mid, rc, 0)
        t3 = time.perf_counter()
        if keep is not None and frames == 0:
            keep.update(input=img.astype(np.float16), want=out, want_mid=mid, checker=o.kind)
        t_easu += t1 - t0
        t_rcas += t3 - t2
        frames += 1
        if time.perf_counter() - t_begin >= target_seconds or frames >= 64:
            break
    mpix = frames * out_h * out_w / 1e6
    return {
        "value": round(mpix / (t_easu + t_rcas), 3), "unit": "Mpix/s", "cores": int(o.threads), "kind": o.kind,
        "sample": "%d whole %dx%d->%dx%d frame(s), FsrEasuF then FsrRcasF (fp32 arithmetic), OpenMP over output rows; "
                  "easu %.2f s + rcas %.2f s of compute" % (frames, in_w, in_h, out_w, out_h, t_easu, t_rcas),
        "easu_mpix_s": round(mpix / t_easu, 3), "rcas_mpix_s": round(mpix / t_rcas, 3),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--regions", type=int, default=7,
                    help="timed regions of exactly --steps steps each (barrier + synchronize either side); the line reports the median region")
    ap.add_argument("--streams", type=int, default=3,
                    help="HIP streams per GPU the steps alternate over (fsr1_pipeline: step i on stream i mod S, each stream with its own "
                         "intermediary, so the tail of one step overlaps the head of the next); 1 = every step on one in-order stream, "
                         "the method of rounds 1-3, which the line also reports as config.one_stream")
    ap.add_argument("--workload", default="1080p_to_4k", choices=sorted(WORKLOADS))
    ap.add_argument("--pipeline", default="two-pass", choices=["two-pass", "fused", "easu", "color"],
                    help="color: the stand-alone colour pass (needs --stages) on an output-sized image")
    ap.add_argument("--stages", type=int, default=0,
                    help="FSR1_COLOR_* bits fused into the pipeline (SURVEY 8f-N4): 1 SRTM prologue, 2 film grain, 4 SRTM inverse, "
                         "8 / 16 TEPD 8-bit / 10-bit dither; e.g. 7 = the HDR chain, 10 = grain + 8-bit dither")
    ap.add_argument("--math", default="strict", choices=["strict", "f", "exact", "h"],
                    help="strict (default, round 6): FSR1_FLAG_MATH_STRICT — fp32 arithmetic, EASU bit-identical to FsrEasuF, the final image within 1 "
                         "binary16 ULP of the reference chain (north_star's tolerance, end to end); f: the faster default arithmetic (per stage <= 1 ULP; "
                         "image level 99.99 %% within 1 ULP, max 6-9: the headline of rounds 1-5); exact: the reference's operation order (bit-identical); "
                         "h: packed fp16 (FsrEasuH / FsrRcasH)")
    ap.add_argument("--storage", default="rgba16f", choices=["rgba16f", "rgba8", "rgba32f"],
                    help="image format in HBM: rgba16f (BASELINE's 8 B/pixel), rgba8 (UNORM, 4 B/pixel; SURVEY 8f-N2) or rgba32f "
                         "(16 B/pixel: BASELINE configs[0], 'fp32 FsrEasuF' with the sample's SAMPLE_SLOW_FALLBACK view, FSR_Pass.glsl:40)")
    ap.add_argument("--graph", type=int, default=0,
                    help="capture this many consecutive steps into one hipGraph and replay it (launch-bound small frames, SURVEY H9); "
                         "--steps is rounded down to a multiple of it")
    ap.add_argument("--ring", type=int, default=0, help="distinct input / output sets to rotate over (0 = enough to exceed 1 GiB)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-cold-rcas", action="store_true",
                    help="skip the extra RCAS launches on an HBM-cold image (profiling runs: keeps rocprofv3's per-kernel average to the pipeline's own launches)")
    ap.add_argument("--no-also", action="store_true",
                    help="skip the also_measured pipelines (profiling runs: only the workload's own kernels in the trace)")
    ap.add_argument("--also", action="store_true",
                    help="with --gpus N > 1: run the also_measured pipelines as well (default at N > 1: headline regions only, so that no "
                         "collective runs outside them and one slow rank cannot distort five more sections)")
    ap.add_argument("--no-pin", action="store_true", help="do not pin the rank's process to the CPUs local to its GPU's PCIe root (sysfs local_cpulist)")
    ap.add_argument("--submit-only", action="store_true",
                    help="measure only the host's submission ceiling (steps/s through fsr1_pipeline_upscale on 16x16 frames) on every rank and print it; "
                         "with --gpus 8 --backend gloo --oversubscribe: eight processes contending on one host")
    ap.add_argument("--no-submit-ceiling", action="store_true", help="skip the host submission ceiling block (profiling runs: its 16x16 launches would sit in the trace)")
    ap.add_argument("--no-telemetry-window", action="store_true",
                    help="skip the dedicated 1.2 s load window the per-rank clock / power figures are sampled in (the line then marks them unreliable)")
    ap.add_argument("--no-latency", action="store_true", help="skip the single-frame latency block (N = 1 only)")
    ap.add_argument("--no-steady", action="store_true", help="skip the steady-state windows (K steps timed INSIDE a longer stream, fill / drain excluded)")
    ap.add_argument("--no-parity", action="store_true", help="skip the image-level parity block (N = 1 only; part of the cpu_baseline leg)")
    ap.add_argument("--no-fast-paths", action="store_true", help="FSR1_FLAG_NO_FAST_PATHS: generic kernels only (A/B of the exact-2x variants)")
    ap.add_argument("--rotate-intermediary", action="store_true",
                    help="two-pass: rotate the EASU->RCAS intermediary over the ring as well (round 1's method; default: one reused buffer, as the sample has)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="process group of the counters-only collectives: nccl (= RCCL over xGMI, the product path) or gloo (host; for boxes "
                         "with fewer GPUs than ranks, see --oversubscribe)")
    ap.add_argument("--oversubscribe", action="store_true",
                    help="with --backend gloo: let ranks share GPUs (rank r runs on device r %% visible devices) — exercises the N-rank launch on a one-GPU box; "
                         "the line says so and its value is not a scaling figure")
    ap.add_argument("--collective", default="auto", choices=["auto", "always"],
                    help="auto: a process group only when there is more than one rank; always: also at N = 1, so that RCCL's communicator, "
                         "barrier, all-reduce and all-gather run on a one-GPU box exactly as they do at N > 1")
    ap.add_argument("--stub", action="store_true", help="test fixture: the N-rank plumbing over gloo with no GPU work (data = 'stub')")
    args = ap.parse_args()

    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.regions < 1:
        raise SystemExit("--regions must be >= 1")
    if not 1 <= args.streams <= 8:
        raise SystemExit("--streams must be 1 .. 8")
    if args.pipeline == "color" or args.rotate_intermediary:
        args.streams = 1  # (the colour pass is not an upscale; a rotated intermediary is a single-stream experiment)
    if os.environ.get("WORLD_SIZE", "1") == "1" and args.gpus > 1 and os.environ.get("FSR1_BENCH_SELF_LAUNCHED") != "1":
        # started like `--gpus 1` is: one bare python process.  Start the ranks ourselves, exactly as the other launch style does.
        raise SystemExit(self_launch(sys.argv[1:], args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("WORLD_SIZE=%d does not match --gpus %d" % (world, args.gpus))
    if args.stub:
        return stub_main(args, world, rank)

    import numpy as np
    import torch
    import torch.distributed as dist

    fsr = importlib.import_module("fidelityfx-fsr_amd")
    fsr.load()  # raises if the HIP library is not built: no fallback

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the FSR1 HIP path has no CPU fallback")
    n_dev = torch.cuda.device_count()
    if local_rank >= n_dev and not (args.oversubscribe and args.backend == "gloo"):
        raise SystemExit("rank %d has no GPU: %d visible device(s) for --gpus %d (one process per GPU; "
                         "--backend gloo --oversubscribe shares devices for a plumbing run)" % (local_rank, n_dev, args.gpus))
    dev_index = local_rank % n_dev
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    grouped = world > 1 or args.collective == "always"
    if grouped:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1 and "MASTER_PORT" not in os.environ:  # a bare `python bench.py --collective always`
            os.environ.update(MASTER_PORT=str(free_port()), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=device)  # "nccl" is RCCL on ROCm
        else:
            dist.init_process_group("gloo")
    coll_device = device if args.backend == "nccl" else torch.device("cpu")

    tele = Telemetry(torch, dev_index)  # (the sampling thread runs from here on; figures are taken from labelled spans only)
    cpu_binding = bind_to_gpu_cpus(tele, enabled=not args.no_pin)
    cpu_binding_before = cpu_binding.pop("_before", None) or os.sched_getaffinity(0)
    if args.submit_only:
        # the host's submission ceiling alone: every rank at the same time (a barrier in front), so that N ranks on one host contend as they
        # would in a run; one JSON line with every rank's rate
        if grouped:
            dist.barrier()
        mine = submission_ceiling(fsr, torch, device, args.streams, seconds=1.0)
        rows = gather_row([mine["steps_per_s"], mine["us_per_step"]], coll_device)
        tele.close()
        if rank == 0:
            need = 1e3 / 0.0610  # steps/s a rank submits at the headline's ~61 us per step
            print(json.dumps({"metric": "host submission ceiling (steps/s per rank through fsr1_pipeline_upscale)", "value": min(r[0] for r in rows), "unit": "steps/s",
                              "n_gpus": world, "per_rank_steps_per_s": [r[0] for r in rows], "per_rank_us_per_step": [r[1] for r in rows],
                              "needed_steps_per_s": round(need, 0), "margin": round(min(r[0] for r in rows) / need, 2), "what": mine["what"],
                              "streams": args.streams, "oversubscribed": bool(world > n_dev), "cpu_binding_rank0": cpu_binding, "data": "synthetic", "higher_is_better": True}), flush=True)
        if grouped:
            dist.destroy_process_group()
        return

    in_w, in_h, out_w, out_h, frames = WORKLOADS[args.workload]
    math_flags = {"f": 0, "strict": fsr.FLAG_MATH_STRICT, "exact": fsr.FLAG_MATH_EXACT, "h": fsr.FLAG_MATH_PACKED_FP16}[args.math]
    if args.no_fast_paths:
        math_flags |= fsr.FLAG_NO_FAST_PATHS
    px = {"rgba16f": 8, "rgba8": 4, "rgba32f": 16}[args.storage]
    tdtype = {"rgba16f": torch.float16, "rgba8": torch.uint8, "rgba32f": torch.float32}[args.storage]
    if args.storage != "rgba16f" and args.math == "h":
        raise SystemExit("--math h (FsrEasuH/FsrRcasH) is defined on RGBA16F images")
    in_bytes, out_bytes = in_w * in_h * px * frames, out_w * out_h * px * frames
    # Buffer layout of a step: inputs and outputs rotate over `ring` sets that together exceed 1 GiB — four times the 256 MB
    # Infinity Cache, so every step reads its input from HBM and writes its output to HBM — while the EASU->RCAS intermediary
    # is ONE buffer reused by every step, as the sample's single intermediary texture is (FSR_Filter.cpp:70-86): for one 4K
    # frame it stays in the Infinity Cache between the two passes and between steps (DESIGN.md section 5 has the sensitivity:
    # rotating the intermediary as well costs 2-3 %).
    # (at least one set more than there are streams: the outputs of steps that may be in flight together must not alias)
    ring = max(args.ring or max(2, -(-(1 << 30) // (in_bytes + out_bytes))), args.streams + 1 if args.streams > 1 else 1)
    # ... and a multiple of the stream count: step i runs on slot i mod S, so set i mod ring is then always reused by a submission on
    # the SAME slot — the only reuse the pipeline orders (include/fsr1_hip.h, "Ordering and aliasing")
    if args.streams > 1:
        ring = -(-ring // args.streams) * args.streams

    # synthetic frames: a few distinct numpy frames uploaded once, then varied on-device per ring slot
    def upload(k):
        f = fsr.frames.synthetic_frame(in_w, in_h, k=k + 16 * rank)
        if args.storage == "rgba8":
            f = np.floor(np.clip(f.astype(np.float32), 0.0, 1.0) * 255.0 + 0.5).astype(np.uint8)
        elif args.storage == "rgba32f":
            f = f.astype(np.float32)
        return torch.from_numpy(f).to(device)

    base = [upload(k) for k in range(min(frames, 2))]
    srcs, dsts = [], []
    for s in range(ring):
        t = torch.stack([torch.roll(base[f % len(base)], shifts=(3 * s + f, 5 * s + 2 * f), dims=(0, 1)) for f in range(frames)])
        srcs.append(t.contiguous())
        dsts.append(torch.empty(frames, out_h, out_w, 4, dtype=tdtype, device=device))
    mids = [torch.empty(frames, out_h, out_w, 4, dtype=tdtype, device=device) for _ in range(ring if args.rotate_intermediary else 1)] \
        if args.pipeline == "two-pass" else [None]
    mid = mids[0]
    stages = None
    if args.stages:
        g = torch.Generator(device="cpu").manual_seed(1234)
        noise = (torch.rand(4, 128, 128, 4, generator=g) - torch.tensor([0.5, 0.5, 0.5, 0.0])).to(torch.float16).to(device)
        stages = fsr.ColorStages(args.stages, grain_amount=0.25, frame=1, noise=noise)  # 128x128 tiles like the sample's blue noise
    if args.pipeline == "color":
        if not stages:
            raise SystemExit("--pipeline color needs --stages")
        srcs = [torch.rand(frames, out_h, out_w, 4, device=device).to(tdtype) if tdtype != torch.uint8 else
                torch.randint(0, 256, (frames, out_h, out_w, 4), device=device, dtype=torch.uint8) for _ in range(ring)]
    easu_con = fsr.FsrEasuCon(in_w, in_h, in_w, in_h, out_w, out_h)
    rcas_con = fsr.FsrRcasCon(0.25)  # sample default attenuation (SampleRenderer.h:49)

    # two-pass: the prologue belongs to EASU's loads, the epilogue to RCAS's stores
    pre = fsr.ColorStages(args.stages & 1) if args.stages & 1 else None
    post = fsr.ColorStages(args.stages & ~1, grain_amount=0.25, frame=1, noise=stages.noise) if args.stages & ~1 else None

    def step(i):
        s = i % ring
        if args.pipeline == "two-pass":
            m = mids[s % len(mids)]
            fsr.easu(srcs[s], m, con=easu_con, flags=math_flags, stages=pre)
            fsr.rcas(m, dsts[s], con=rcas_con, flags=math_flags, stages=post)
        elif args.pipeline == "fused":
            fsr.easu_rcas_fused(srcs[s], dsts[s], easu_con=easu_con, rcas_con=rcas_con, flags=math_flags, stages=stages)
        elif args.pipeline == "color":
            fsr.color(srcs[s], dsts[s], stages, flags=math_flags)
        else:  # EASU only: its output is the pipeline's last image (what fsr1_upscale does with use_rcas = 0)
            fsr.easu(srcs[s], dsts[s], con=easu_con, flags=math_flags | fsr.FLAG_OUTPUT_STREAMING, stages=stages)

    # --streams S > 1: the same step through an fsr1_pipeline — step i on stream i mod S, that stream's own intermediary
    # (managed=False: the bare C ABI — the buffers live for the whole run and every timed region is bracketed by synchronizes)
    pipe = fsr.Pipeline(args.streams, managed=False) if args.streams > 1 else None
    if pipe is not None and args.pipeline == "two-pass":
        # no allocation inside a timed region or a graph capture (a batch whose per-frame intermediaries fit the Infinity Cache together is
        # submitted frame by frame, fsr1_pipeline_upscale: one frame's worth per slot is then enough)
        split = frames > 1 and (out_bytes // frames) * args.streams <= (256 << 20)
        pipe.reserve(out_bytes // frames if split else out_bytes)

    def piped(flags, fused, use_rcas=True, inputs=None, con_stages=None):
        def fn(i):
            s_ = i % ring
            pipe.upscale((inputs or srcs)[s_], dsts[s_], sharpness=0.25, use_rcas=use_rcas, fused=fused, flags=flags, stages=con_stages)
        return fn

    step1 = step  # the one-stream form
    if pipe is not None:
        step = piped(math_flags, 1 if args.pipeline == "fused" else 0, use_rcas=args.pipeline != "easu", con_stages=stages)

    def fence():
        torch.cuda.synchronize()
        if grouped:
            dist.barrier()
        torch.cuda.synchronize()

    def close():
        """Closing bracket of a timed region: synchronize, read this rank's clock, then barrier + synchronize.  The
        region is bracketed by a barrier and a synchronize on both sides; the clock is read once this rank's K steps have
        left the device and before the closing barrier's own latency (a few tens of microseconds of RCCL launch, which
        at the driver's K = 20 would be several percent of the region) — the MAX over ranks, taken by reduce_counters,
        is then the time at which the slowest rank finished its K steps."""
        torch.cuda.synchronize()
        t = time.perf_counter()
        if grouped:
            dist.barrier()
            torch.cuda.synchronize()
        return t

    # Device clock ramp (not part of W): an idle MI355X sits at ~100 MHz and needs a few milliseconds of load
    # to reach its working clocks; EASU then runs AT the 1400 W package power cap (sclk ~2.1 of 2.4 GHz), so the
    # steady state is what a frame stream sees.  ~0.2 s of the same steps, untimed.
    viol0 = tele.violations(dev_index) if not args.no_telemetry_window else None  # (before the ramp: the GPU is idle anyway)
    t_ramp = time.perf_counter()
    i = 0
    while time.perf_counter() - t_ramp < 0.2:
        step(i)
        i += 1
        if i % 64 == 0:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    # Telemetry window (not part of W, outside every timed region): TELEMETRY_WINDOW_S of the headline's own pipelined load, sampled every 2 ms —
    # 500+ samples whatever --steps is (round 5 sampled across the run itself: at the driver's K = 20 that was 57 samples of a 0.1 s load and
    # read 20 % low).  The SMU's throttle accumulators are read before and after it: their difference says which limiter was active.
    if not args.no_telemetry_window:
        tele.begin("headline")
        t_win = time.perf_counter()
        while time.perf_counter() - t_win < TELEMETRY_WINDOW_S:
            step(i)
            i += 1
            if i % 64 == 0:
                torch.cuda.synchronize()
        torch.cuda.synchronize()
        tele.end("headline")
    viol1 = tele.violations(dev_index) if viol0 is not None else None
    if viol0 is not None:
        # reading the accumulators took ~1 s of host time with the GPU idle (it drops to ~100 MHz within a millisecond): ramp again, so that
        # the warm-up and the timed regions start from working clocks exactly as they did before the window existed
        t_ramp = time.perf_counter()
        while time.perf_counter() - t_ramp < 0.2:
            step(i)
            i += 1
            if i % 64 == 0:
                torch.cuda.synchronize()
    for i in range(args.warmup):
        step(i)
    fence()

    def timed(fn, k, first=0):
        """R timed regions of EXACTLY k steps each, every one bracketed by barrier + synchronize on both sides (fence /
        close).  Returns (the R region times as the MAX over ranks, this rank's own R times).  The line's `ms_per_step` and
        `value` use the MEDIAN region: at the driver's K = 20 a region is 1.3 ms, and one such sample — at N = 8 the MAX over
        eight ranks of one such sample — is decided by whichever rank's first launch was slowest."""
        own = []
        for r in range(args.regions):
            fence()
            t0 = time.perf_counter()
            for i in range(k):
                fn(first + r * k + i)
            own.append(close() - t0)
        return reduce_regions(own, coll_device), own

    if args.graph > 0:
        # K steps = K / graph replays of a hipGraph holding `graph` consecutive steps (ring slots 0 .. graph-1, ...)
        args.steps = max(args.graph, args.steps // args.graph * args.graph)
        g = torch.cuda.CUDAGraph()
        cap = torch.cuda.Stream()
        with torch.cuda.stream(cap):
            with torch.cuda.graph(g, stream=cap):
                if pipe is not None:
                    pipe.fork(cap)  # the pipeline's streams join the capture: the graph holds `streams` parallel chains of steps
                for i in range(args.graph):
                    step(i)
                if pipe is not None:
                    pipe.join(cap)
        g.replay()
        regions, own_regions = timed(lambda i: g.replay(), args.steps // args.graph)
    else:
        regions, own_regions = timed(step, args.steps, first=args.warmup)
    seconds = median(regions)

    def steady(fn, k, stream_of_last):
        """Steady state: R windows of exactly k steps timed INSIDE one longer run — `lead` steps in front (the pipeline fills), R x k
        steps, `lead` steps behind (it drains), one fence either side of the whole run and none inside.  Window r is the time from
        the completion of the step in front of it to the completion of its own last step, taken with HIP events recorded on the
        streams those steps run on (`stream_of_last()` = the stream the latest submission went to).  What a K-step region adds to
        this — the first launch's latency and the drain of the last frames, 1-5 % at K = 20 — is what the two figures differ by.
        Returns the R window times (seconds, MAX over ranks)."""
        lead = max(k, 4 * args.streams, 8)
        timers = [fsr.Timer() for _ in range(args.regions)]
        fence()
        n = args.warmup
        for _ in range(lead):
            fn(n)
            n += 1
        for t in timers:
            t.start(stream_of_last())
            for _ in range(k):
                fn(n)
                n += 1
            t.stop(stream_of_last())
        for _ in range(lead):
            fn(n)
            n += 1
        close()
        return reduce_regions([t.elapsed_ms() * 1e-3 for t in timers], coll_device)

    def last_pipe_stream():
        return pipe._slot_stream((pipe.next_slot() - 1) % pipe.streams)

    steady_regions = None
    if not args.no_steady and args.graph == 0:
        steady_regions = steady(step, args.steps, last_pipe_stream if pipe is not None else (lambda: None))

    # the same K steps on ONE in-order stream (the method of rounds 1-3): the per-kernel stopwatch below is taken that way, and the
    # line's self-consistency check compares the kernels with THIS step (overlapped steps are shorter than the kernels they contain)
    tele.begin("one_stream")
    one_regions = timed(step1, args.steps, first=args.warmup)[0] if pipe is not None else regions
    one_steady = steady(step1, args.steps, lambda: None) if (pipe is not None and steady_regions is not None) else steady_regions
    tele.end("one_stream")

    total = reduce_counters(frames * args.steps, frames * args.steps * out_w * out_h, median(own_regions), coll_device)
    total["seconds"] = seconds  # median over regions of the max over ranks
    per_rank_seconds = gather_seconds(median(own_regions), coll_device)
    value = total["pixels"] / total["seconds"] / 1e6

    # ---- per-kernel durations with HIP events on the launch stream (C-ABI stopwatch) ----
    # Taken right after the headline regions (same clocks, same buffers) and INDEPENDENT of --steps: after a ramp of the very
    # kernel being timed (>= 20 launches and >= 30 ms, so that clocks and power state are the steady ones — an idle MI355X drops
    # its clock within a millisecond, and round 3's 20-launch stopwatch read 1.2-1.4 x the rocprofv3 average for that reason),
    # B blocks of n launches each between one pair of events; the figure is the MEDIAN block.
    timer = fsr.Timer()

    def kernel_ms(fn, blocks=5):
        t_r, i = time.perf_counter(), 0
        while i < 20 or time.perf_counter() - t_r < 0.03:
            fn(i)
            i += 1
            if i % 64 == 0:
                torch.cuda.synchronize()
        torch.cuda.synchronize()
        est = max((time.perf_counter() - t_r) / i, 1e-6)  # host-clocked seconds per launch of the ramp
        n = max(20, min(200, int(0.25 / est)))              # 200 launches for a 4K frame; fewer for multi-millisecond batches
        per = []
        for _ in range(blocks):
            timer.start()
            for j in range(n):
                fn(i + j)
            timer.stop()
            per.append(timer.elapsed_ms() / n)
            i += n
        return median(per), {"launches_per_block": n, "blocks": blocks, "min_us": round(min(per) * 1e3, 2), "max_us": round(max(per) * 1e3, 2)}

    kern, kern_info = {}, {}
    rcas_cold_ms = None

    def stopwatch(name, fn):
        kern[name], kern_info[name] = kernel_ms(fn)

    if args.pipeline in ("two-pass", "easu"):
        eflags = math_flags | (0 if args.pipeline == "two-pass" else fsr.FLAG_OUTPUT_STREAMING)
        stopwatch("easu", lambda i: fsr.easu(srcs[i % ring], mid if args.pipeline == "two-pass" else dsts[i % ring], con=easu_con,
                                             flags=eflags, stages=pre if args.pipeline == "two-pass" else stages))
    if args.pipeline == "two-pass":
        # as inside the pipeline: the input is the intermediary EASU left behind (for one 4K frame, in the Infinity Cache)
        stopwatch("rcas", lambda i: fsr.rcas(mid, dsts[i % ring], con=rcas_con, flags=math_flags, stages=post))
        if not args.stages and ring >= 4 and not args.no_cold_rcas:
            # and on an image that comes from HBM: an output written ring/2 steps ago (non-temporal stores, > 512 MB of traffic since)
            rcas_cold_ms = kernel_ms(lambda i: fsr.rcas(dsts[(i + ring // 2) % ring], dsts[i % ring], con=rcas_con, flags=math_flags))[0]
    if args.pipeline == "fused":
        stopwatch("fused", lambda i: fsr.easu_rcas_fused(srcs[i % ring], dsts[i % ring], easu_con=easu_con, rcas_con=rcas_con,
                                                         flags=math_flags, stages=stages))
    if args.pipeline == "color":
        stopwatch("color", lambda i: fsr.color(srcs[i % ring], dsts[i % ring], stages, flags=math_flags))

    # The single-launch pipeline (BASELINE configs[3]) writes the very same image as the two dispatches
    # (tests: fused == two-pass, bit for bit); the default run times it over the same K steps (and R regions) as well and
    # reports it beside the headline, which stays the two-dispatch pipeline BASELINE's metric is quoted on.
    also = None

    def also_entry(fn1, note, px_per_step=None, fn=None):
        """fn1: the step on one in-order stream; fn: the same step through the pipeline (--streams > 1)"""
        px = (px_per_step or frames * out_w * out_h) * args.steps * world

        def measure(f):
            for i in range(min(args.warmup, 50)):
                f(i)
            reg = timed(f, args.steps)[0]
            sec = median(reg)
            return {"value": round(px / sec / 1e6, 1), "unit": "Mpix/s", "ms_per_step": round(sec * 1e3 / args.steps, 5),
                    "ms_per_step_min": round(min(reg) * 1e3 / args.steps, 5), "ms_per_step_max": round(max(reg) * 1e3 / args.steps, 5)}
        one = measure(fn1)
        if pipe is None or fn is None:
            return dict(one, streams=1, note=note)
        return dict(measure(fn), streams=args.streams, one_stream={k: one[k] for k in ("value", "ms_per_step")}, note=note)

    run_also = not args.no_also and (world == 1 or args.also)  # at N > 1 only the headline regions hold collectives, unless --also
    if args.pipeline == "two-pass" and not args.stages and args.math != "h" and run_also:
        def fused_step(i):
            fsr.easu_rcas_fused(srcs[i % ring], dsts[i % ring], easu_con=easu_con, rcas_con=rcas_con, flags=math_flags)
        also = {"fused": also_entry(fused_step, "EASU->RCAS in one launch, output bit-identical to the two dispatches (BASELINE configs[3])",
                                    fn=pipe and piped(math_flags, 1))}
        # the single launch as a kernel: HIP-event time per launch, its algorithmic bytes (in + out) against the HBM line, and
        # the PMC traffic / VALU count of its own committed profile when that was taken of the running sources
        tf_ms, tf_info = kernel_ms(fused_step)
        fpmc = pmc_traffic(fsr, args.workload, "fused", "fused", args.math, args.storage) if not args.no_fast_paths else (None, None, None, None)
        also["fused"].update({"avg_kernel_us": round(tf_ms * 1e3, 2), "stopwatch": tf_info, "algorithmic_bytes": in_bytes + out_bytes,
                              "hbm_frac": round((in_bytes + out_bytes) / (tf_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                              "traffic": fpmc[0], "traffic_source": fpmc[1]})
        if fpmc[2]:
            also["fused"]["valu_frac"] = round(fpmc[2] / (tf_ms * 1e-3) / 1e9 / VALU_PEAK_GINST, 4)
        if args.math == "strict":
            # the faster default arithmetic (rounds 1-5's headline: per stage within 1 binary16 ULP of FsrEasuF / FsrRcasF, at image level 99.99 %
            # of the values within 1 ULP of the reference chain, max 6 at 0.25 stops) on the same frames and K steps: what F-strict's guarantee costs
            def d_step(i):
                fsr.easu(srcs[i % ring], mid, con=easu_con, flags=0)
                fsr.rcas(mid, dsts[i % ring], con=rcas_con, flags=0)

            def df_step(i):
                fsr.easu_rcas_fused(srcs[i % ring], dsts[i % ring], easu_con=easu_con, rcas_con=rcas_con, flags=0)
            also["default_arithmetic_two_pass"] = also_entry(d_step, "the default (F) arithmetic, two dispatches: per stage <= 1 binary16 ULP, image level 99.99 % within 1 ULP / max 6 "
                                                                     "(the headline of rounds 1-5; `parity.default_arithmetic` has this run's histogram)", fn=pipe and piped(0, 0))
            also["default_arithmetic_fused"] = also_entry(df_step, "the default (F) arithmetic, EASU->RCAS in one launch", fn=pipe and piped(0, 1))
            # ... and the default arithmetic's EASU as a kernel (the dominant kernel of rounds 1-5's headline), for continuity of the roofline figure
            de_ms, de_info = kernel_ms(lambda i: fsr.easu(srcs[i % ring], mid, con=easu_con, flags=0))
            also["default_arithmetic_two_pass"]["easu_kernel"] = {"avg_kernel_us": round(de_ms * 1e3, 2), "stopwatch": de_info, "algorithmic_bytes": in_bytes + out_bytes,
                                                                  "hbm_frac": round((in_bytes + out_bytes) / (de_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)}
        if args.storage == "rgba16f" and args.math in ("f", "strict"):
            # the reference's shipping default (FsrEasuH / FsrRcasH, FSR_Pass.hlsl:81-87) on the same frames, same K steps
            hflags = fsr.FLAG_MATH_PACKED_FP16

            def h_step(i):
                fsr.easu(srcs[i % ring], mid, con=easu_con, flags=hflags)
                fsr.rcas(mid, dsts[i % ring], con=rcas_con, flags=hflags)
            # and the bit-exact arithmetic (FSR1_FLAG_MATH_EXACT: the reference's operation order, no re-association): what
            # bit-identity with the CPU-evaluated FsrEasuF / FsrRcasF costs against the default (<= 1 ULP) arithmetic
            eflags_x = fsr.FLAG_MATH_EXACT

            def x_step(i):
                fsr.easu(srcs[i % ring], mid, con=easu_con, flags=eflags_x)
                fsr.rcas(mid, dsts[i % ring], con=rcas_con, flags=eflags_x)
            also["exact_two_pass"] = also_entry(x_step, "FSR1_FLAG_MATH_EXACT: bit-identical to the CPU-evaluated FsrEasuF + FsrRcasF (0 differing values on whole frames)",
                                                fn=pipe and piped(eflags_x, 0))
            also["packed_fp16_two_pass"] = also_entry(h_step, "FsrEasuH + FsrRcasH (parity class H: bit-exact vs the reference's packed-fp16 path; "
                                                              "v_pk_*_f16 issue at half rate on MI355X, DESIGN.md 3.4)", fn=pipe and piped(hflags, 0))

            def hf_step(i):
                fsr.easu_rcas_fused(srcs[i % ring], dsts[i % ring], easu_con=easu_con, rcas_con=rcas_con, flags=hflags)
            also["packed_fp16_fused"] = also_entry(hf_step, "FsrEasuH -> FsrRcasH in one launch, bit-identical to the two H dispatches", fn=pipe and piped(hflags, 1))

        if args.workload == "1080p_to_4k" and args.storage == "rgba16f" and args.math in ("f", "strict") and not args.no_fast_paths:
            # BASELINE configs[2]'s shape on the same box and K steps (one 2560x1440 -> 3840x2160 frame per step, "Quality"
            # 1.5x): the ratios without a quad form run the generic kernels, which the exact-2x headline never touches
            q_in = [torch.roll(torch.from_numpy(fsr.frames.synthetic_frame(2560, 1440, k=3 + 16 * rank)).to(device), shifts=(3 * s, 5 * s), dims=(0, 1)).contiguous().unsqueeze(0)
                    for s in range(ring)]
            q_con = fsr.FsrEasuCon(2560, 1440, 2560, 1440, out_w, out_h)

            def q_step(i):
                fsr.easu(q_in[i % ring], mid, con=q_con, flags=math_flags)
                fsr.rcas(mid, dsts[i % ring], con=rcas_con, flags=math_flags)
            also["quality_1440p_to_4k_two_pass"] = also_entry(q_step, "2560x1440 -> 3840x2160 (1.5x, BASELINE configs[2]'s shape), EASU + RCAS as two dispatches, generic kernels",
                                                              px_per_step=out_w * out_h, fn=pipe and piped(math_flags, 0, inputs=q_in))
            del q_in
            # ... and the reference sample's true-ratio presets at a 4K target (sample/src/DX12/FSRSample.h:79-95, PDF p.10): "Ultra Quality"
            # 1.3x and "Balanced" 1.7x — with 1.5x above, three of its four quality modes run the generic kernel
            for key, (p_w, p_h, label) in {"ultra_quality_1662p_to_4k_two_pass": (2954, 1662, "1.3x 'Ultra Quality'"),
                                           "balanced_1270p_to_4k_two_pass": (2259, 1270, "1.7x 'Balanced'")}.items():
                p_in = [torch.roll(torch.from_numpy(fsr.frames.synthetic_frame(p_w, p_h, k=4 + 16 * rank)).to(device), shifts=(3 * s, 5 * s), dims=(0, 1)).contiguous().unsqueeze(0)
                        for s in range(ring)]
                p_con = fsr.FsrEasuCon(p_w, p_h, p_w, p_h, out_w, out_h)

                def p_step(i, p_in=p_in, p_con=p_con):
                    fsr.easu(p_in[i % ring], mid, con=p_con, flags=math_flags)
                    fsr.rcas(mid, dsts[i % ring], con=rcas_con, flags=math_flags)
                also[key] = also_entry(p_step, "%dx%d -> 3840x2160 (%s, the sample's true-ratio preset), EASU + RCAS as two dispatches, generic kernels" % (p_w, p_h, label),
                                       px_per_step=out_w * out_h, fn=pipe and piped(math_flags, 0, inputs=p_in))
                del p_in

    # ---- single-frame latency (N = 1): submit -> completion of ONE frame on a warm, otherwise idle GPU — the reference's actual
    #      usage, one Upscale per display refresh (SampleRenderer.cpp:705-709), which the pipelined headline does not represent ----
    latency = None
    if world == 1 and not args.no_latency and args.pipeline in ("two-pass", "fused") and not args.stages and frames == 1:
        def lat(submit, n=200, gap=1e-3):
            for i in range(50):
                submit(i)
            torch.cuda.synchronize()
            host, devt = [], []
            for i in range(n):
                if gap:
                    time.sleep(gap)  # the GPU idles (and starts to drop its clock) like between two display refreshes
                timer.start()
                t0 = time.perf_counter()
                submit(i)
                timer.stop()
                torch.cuda.current_stream().synchronize()
                host.append(time.perf_counter() - t0)
                devt.append(timer.elapsed_ms() * 1e-3)
            host.sort()
            return {"host_us": round(median(host) * 1e6, 1), "host_p90_us": round(host[int(0.9 * (n - 1))] * 1e6, 1), "device_us": round(median(devt) * 1e6, 1)}

        lat_mid = mid if mid is not None else torch.empty_like(dsts[0])

        def two_pass_1(i):
            fsr.easu(srcs[i % ring], lat_mid, con=easu_con, flags=math_flags)
            fsr.rcas(lat_mid, dsts[i % ring], con=rcas_con, flags=math_flags)

        def fused_1(i):
            fsr.easu_rcas_fused(srcs[i % ring], dsts[i % ring], easu_con=easu_con, rcas_con=rcas_con, flags=math_flags)
        filt = fsr.FSR_Filter()
        filt.OnCreate(slowFallback=args.math != "h", exact=args.math == "exact", fused="auto", strict=args.math == "strict")
        filt.OnCreateWindowSizeDependentResources(srcs[0], dsts[0], out_w, out_h)
        auto_state = fsr.State(in_w, in_h, bUseRcas=True, rcasAttenuation=0.25)
        modes = {"two_pass": two_pass_1, "fused": fused_1, "auto": lambda i: filt.Upscale(out_w, out_h, auto_state)}
        tele.begin("latency")
        latency = {"frames_per_mode": 200, "idle_gap_ms": 1.0,
                   "definition": "host_us: time.perf_counter() around submit + hipStreamSynchronize of ONE frame (what an integrator waits), median and p90 of 200 "
                                 "frames with 1 ms of idle GPU before each; device_us: HIP events on the stream just before / after the frame's launches; "
                                 "back_to_back: the same without the idle gap (the GPU keeps its clocks)",
                   "after_1ms_idle": {k: lat(f) for k, f in modes.items()},
                   "back_to_back": {k: lat(f, gap=0.0) for k, f in modes.items()}}
        tele.end("latency")
        filt.OnDestroy()

    # the host's submission ceiling (outside every timed region): steps/s this rank's thread can push through the pipeline — every rank at
    # the same time, so that N ranks on one host contend as they do in the timed regions — against what the headline needs
    submit = submit_rows = None
    if not args.no_submit_ceiling:
        if grouped:
            dist.barrier()
        submit = submission_ceiling(fsr, torch, device, args.streams)
        submit_rows = gather_row([submit["steps_per_s"]], coll_device)

    # per-rank shader clock and package power during the headline regions (sysfs, best effort) — one counters-only collective
    tele.close()
    head_mhz, head_w, head_n = tele.summary("headline")

    def viol_delta(*names):
        """after - before of the first violation accumulator whose name contains one of `names` (None when not reported)"""
        if not viol0 or not viol1:
            return None
        for k in viol1:
            if any(n in k.lower() for n in names) and "acc" in k.lower() and k in viol0:
                return viol1[k] - viol0[k]
        return None
    tele_rows = gather_row([head_mhz, head_w, head_n, tele.power_cap_w, viol_delta("ppt"), viol_delta("socket_thermal", "thermal")], coll_device)
    per_rank_mhz, per_rank_watts = [r[0] for r in tele_rows], [r[1] for r in tele_rows]

    # algorithmic HBM bytes per launch: EASU in+out, RCAS 2*out, fused in+out
    alg = {"easu": in_bytes + out_bytes, "rcas": 2 * out_bytes, "fused": in_bytes + out_bytes, "color": 2 * out_bytes}
    dominant = max(kern, key=kern.get)

    def roof(name):
        gbps = alg[name] / (kern[name] * 1e-3) / 1e9
        pmc = pmc_traffic(fsr, args.workload, args.pipeline, name, args.math, args.storage) if not args.stages and not args.no_fast_paths else (None, None, None, None)
        r = {"kernel": name, "bound": "hbm", "achieved": round(gbps, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
             "frac": round(gbps / HBM_PEAK_GBPS, 4), "traffic": pmc[0],
             # PMC counters cannot be read inside the timed run: the figure is looked up in the committed rocprofv3 summary of
             # this configuration, and only when that summary was taken of the kernel sources running now (source_hash)
             "traffic_kind": "profile-lookup" if pmc[0] is not None else None,
             "traffic_source": pmc[1], "algorithmic_bytes": alg[name],
             "avg_kernel_us": round(kern[name] * 1e3, 2), "stopwatch": kern_info[name]}
        if name == "rcas":
            r["input"] = "the EASU->RCAS intermediary, one buffer reused by every step (Infinity-Cache resident for a single 4K frame), as inside the pipeline"
            if rcas_cold_ms:
                r["cold_input"] = {"avg_kernel_us": round(rcas_cold_ms * 1e3, 2), "frac": round(alg[name] / (rcas_cold_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                                   "note": "the same pass on an image read from HBM"}
        if pmc[3]:
            r["traffic_stale"] = "%s was taken of other kernel sources (source_hash differs): not quoted" % pmc[3]
        if name in ("easu", "fused"):
            # the north star's "HBM read-roofline": input bytes only (SURVEY.md 8d asks for both figures, labelled)
            rd = in_bytes / (kern[name] * 1e-3) / 1e9
            r["read_only"] = {"achieved": round(rd, 1), "frac": round(rd / HBM_PEAK_GBPS, 4), "algorithmic_bytes": in_bytes}
        if pmc[2]:
            # the wall this kernel is actually on: VALU issue (wave-instructions per launch from the same PMC summary,
            # SQ_INSTS_VALU, over the live kernel time)
            gi = pmc[2] / (kern[name] * 1e-3) / 1e9
            r["valu"] = {"achieved": round(gi, 1), "peak": VALU_PEAK_GINST, "unit": "G wave-inst/s", "frac": round(gi / VALU_PEAK_GINST, 4),
                         "wave_insts_per_launch": int(pmc[2])}
        return r

    if rank == 0:
        line = {
            "metric": {"easu": "upscaled megapixels/sec (EASU only)", "color": "megapixels/sec (colour pass)"}.get(
                args.pipeline, "upscaled megapixels/sec (EASU+RCAS, %s %s)" % ("1080p->4K" if args.workload == "1080p_to_4k" else args.workload,
                                                                                  {"rgba16f": "fp16", "rgba8": "rgba8 unorm", "rgba32f": "fp32 storage"}[args.storage])),
            "value": round(value, 1), "unit": "Mpix/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(total["seconds"] * 1e3 / args.steps, 5), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": (round(value / world / {(3840, 2160): PUBLISHED_4K_MPIX_S, (2560, 1440): PUBLISHED_1440P_MPIX_S}[(out_w, out_h)], 2)
                            if (out_w, out_h) in ((3840, 2160), (2560, 1440)) and args.pipeline in ("two-pass", "fused") and not args.stages else None),
            "vs_baseline_ref": "per GPU, vs BASELINE.md §1: EASU+RCAS <= 0.40 ms per 4K frame (<= 0.20 ms per 1440p frame) on RX 6800 XT / RTX 3080 (reference PDF p.9) = >= 20736 (18432) Mpix/s; "
                               "an upper bound on time, measured on other hardware",
            "dtype": "f32" if args.math != "h" else "f16", "data": "synthetic",
            # which parity class the headline's arithmetic belongs to (include/fsr1_hip.h, FSR1_FLAG_MATH_*):
            #   F-default: fp32 arithmetic, per stage <= 1 binary16 ULP of the CPU-evaluated FsrEasuF / FsrRcasF; image level: see `parity`
            #   EXACT: bit-identical to the reference chain;  H: bit-identical to the CPU-evaluated FsrEasuH / FsrRcasH
            #   F-strict (round 6): fp32 arithmetic, EASU's stored image bit-identical to FsrEasuF's, the final image within 1 binary16 ULP of the chain
            "parity_class": {"strict": "F-strict", "f": "F-default", "exact": "EXACT", "h": "H"}[args.math],
            "config": {"workload": "%s: %dx%d -> %dx%d %s, %d frame(s)/step/GPU, %s, math=%s, ring of %d frame sets"
                                   % (args.workload, in_w, in_h, out_w, out_h, args.storage.upper(), frames, args.pipeline, args.math, ring)
                                   + (" (inputs / outputs; one reused intermediary)" if args.pipeline == "two-pass" else ""),
                       "source_hash": fsr._lib.source_hash(), "build_id": fsr._lib.build_id(),
                       "binary_matches_sources": fsr._lib.build_id() == fsr._lib.source_hash(),
                       "intermediary": None if args.pipeline != "two-pass" else ("one per stream" if pipe is not None else ("rotated" if args.rotate_intermediary else "reused")),
                       "launch": launch_style(), "world_size_seen": dist.get_world_size() if grouped else 1,
                       "cpu_binding": cpu_binding,
                       "collective_backend": (("rccl (torch.distributed 'nccl')" if args.backend == "nccl" else "gloo") if grouped else None),
                       "devices_visible": n_dev, "oversubscribed": bool(world > n_dev),
                       "streams": args.streams,
                       "one_stream": {"value": round(total["pixels"] / median(one_regions) / 1e6, 1), "ms_per_step": round(median(one_regions) * 1e3 / args.steps, 5),
                                      "note": "the same K steps on one in-order HIP stream (the method of rounds 1-3; what the kernels' durations add up to)"},
                       "pipelining": ("step i runs on HIP stream i mod %d (fsr1_pipeline), each stream with its own EASU->RCAS intermediary: frames are independent, so the "
                                      "tail of one step overlaps the head of the next (a kernel boundary costs ~5 us of an otherwise idle chip)" % args.streams)
                                     if pipe is not None else "none: one in-order stream",
                       "regions": args.regions,
                       "region_ms_per_step": {"median": round(median(regions) * 1e3 / args.steps, 5), "min": round(min(regions) * 1e3 / args.steps, 5),
                                              "max": round(max(regions) * 1e3 / args.steps, 5)},
                       "timing": "R = `regions` timed regions of exactly K = `steps` steps, each bracketed by barrier + synchronize on both sides and "
                                 "reduced with MAX over ranks; value / ms_per_step are the MEDIAN region (a K = 20 region of 4K frames is 1.3 ms: the first "
                                 "launch's latency and the closing synchronize are 1-6 % of it, so single regions scatter by that much)",
                       "pipeline": args.pipeline, "storage": args.storage, "color_stages": args.stages, "hip_graph_steps": args.graph, "rcas_sharpness_stops": 0.25,
                       "parallelism": "independent frames per GPU, counters-only collective"},
            # the headline `value` is the pipelined figure (`streams` HIP streams per GPU, the product's way to run a frame stream); the same K
            # steps on ONE in-order stream — the method of rounds 1-3, and what the kernels' durations add up to — at the top level beside it
            "streams": args.streams,
            "one_stream": {"value": round(total["pixels"] / median(one_regions) / 1e6, 1), "ms_per_step": round(median(one_regions) * 1e3 / args.steps, 5)},
            "per_rank_seconds": [round(t, 6) for t in per_rank_seconds],
            # shader clock (median MHz) and package power (mean W) every rank's GPU showed under the headline's load — a dedicated window of
            # TELEMETRY_WINDOW_S of it before the timed regions (`telemetry.window`) — read from the amdgpu driver's sysfs files by a host
            # thread; null where a box does not expose them
            "per_rank_mhz": per_rank_mhz, "per_rank_watts": per_rank_watts,
            "telemetry": telemetry_block(tele_rows, tele.source, tele.period * 1e3, TELEMETRY_WINDOW_S, tele.partition,
                                         extra={"one_stream": dict(zip(("mhz", "watts", "samples"), tele.summary("one_stream"))),
                                                "latency": dict(zip(("mhz", "watts", "samples"), tele.summary("latency"))),
                                                "violation_fields_rank0": sorted(viol1)[:24] if viol1 else None}),
            "host_submission": None if submit is None else dict(
                submit, per_rank_steps_per_s=[r[0] for r in submit_rows], needed_steps_per_s=round(args.steps / total["seconds"], 0),
                margin=round(min(r[0] for r in submit_rows) / (args.steps / total["seconds"]), 2),
                note="margin = the slowest rank's ceiling / the steps per second the headline ran at: well above 1 means the host is not what a rank waits for"),
            "roofline": roof(dominant),
            "kernels": {k: roof(k) for k in kern},
            "pipeline_hbm": {"algorithmic_bytes_per_step": sum(alg[k] for k in kern),
                             "achieved_GBps": round(sum(alg[k] for k in kern) * args.steps / seconds / 1e9, 1)},
        }
        if steady_regions is not None:
            def _st(reg):
                return {"value": round(total["pixels"] / median(reg) / 1e6, 1), "ms_per_step": round(median(reg) * 1e3 / args.steps, 5),
                        "ms_per_step_min": round(min(reg) * 1e3 / args.steps, 5), "ms_per_step_max": round(max(reg) * 1e3 / args.steps, 5)}
            line["steady_state"] = dict(_st(steady_regions), windows=args.regions, one_stream=_st(one_steady),
                                        method="R windows of exactly K steps timed with HIP events INSIDE one longer run (K or more steps in front and behind, no "
                                               "fence inside): the pipeline's fill and drain, which a K-step region includes (1-5 % at K = 20), are excluded")
        if latency:
            line["latency_us"] = latency
        if also:
            line["also_measured"] = also
        elif world > 1 and not args.no_also:
            line["also_measured"] = None
            line["also_measured_note"] = "skipped at n_gpus > 1 (only the headline regions hold collectives); pass --also to run them"
        check_stopwatch(line)  # sum of kernel times <= the step that contains them, or the line says "stopwatch_suspect"
        if world == 1 and not args.no_cpu_baseline and args.pipeline != "color" and not args.stages:
            kept = {}
            try:  # the reference on ALL host cores, as in every round: the rank's pinning to its GPU's NUMA node is lifted for this leg
                if cpu_binding.get("applied"):
                    os.sched_setaffinity(0, cpu_binding_before)
                line["cpu_baseline"] = cpu_baseline(fsr, in_w, in_h, out_w, out_h, keep=kept)
            finally:
                if cpu_binding.get("applied"):
                    os.sched_setaffinity(0, parse_cpulist(cpu_binding["local_cpus"]) & cpu_binding_before)
            if not args.no_parity and kept and args.storage == "rgba16f" and args.math in ("f", "strict", "exact") and args.pipeline in ("two-pass", "fused"):
                # Image-level parity of THIS workload's arithmetic against the reference chain the cpu_baseline leg just evaluated
                # (FsrEasuF -> RTNE binary16 -> FsrRcasF, ffx_fsr1.h:315-437, :684-769): the final image of the two dispatches, the fused
                # launch and a pipelined frame, as a binary16-ULP histogram over R, G, B.  The oracle is the checker here, never the
                # thing measured; the full table over every BASELINE shape and true-ratio preset is tests/test_gpu_image_parity.py ->
                # profiles/r05_image_parity.json.
                import image_parity
                p_src = torch.from_numpy(kept["input"]).to(device)
                p_mid = torch.empty(out_h, out_w, 4, dtype=torch.float16, device=device)
                p_out = {k: torch.empty_like(p_mid) for k in ("two_dispatch", "fused", "pipelined")}
                fsr.easu(p_src, p_mid, con=easu_con, flags=math_flags)
                fsr.rcas(p_mid, p_out["two_dispatch"], con=rcas_con, flags=math_flags)
                fsr.easu_rcas_fused(p_src, p_out["fused"], easu_con=easu_con, rcas_con=rcas_con, flags=math_flags)
                pp = fsr.Pipeline(max(args.streams, 2))
                for _ in range(pp.streams):
                    pp.upscale(p_src, p_out["pipelined"], sharpness=0.25, fused=0 if args.pipeline == "two-pass" else 1, flags=math_flags)
                    pp.synchronize()
                pp.close()
                torch.cuda.synchronize()
                line["parity"] = {"against": "the reference chain FsrEasuF -> RTNE binary16 -> FsrRcasF on the same frame, evaluated by the cpu_baseline leg (kind: %s)" % kept["checker"],
                                  "unit": "binary16 ULP of the final image, R/G/B values",
                                  "gate": "tests/test_gpu_image_parity.py: F-strict max 1 ULP and the intermediary 0 differing values; F-default >= 99.98 % within 1 ULP, "
                                          ">= 99.95 % bit-equal, max 8 (16 at 0 stops); EXACT 0 differing values",
                                  "all_shapes": "profiles/r06_image_parity.json"}
                keys = ("max_ulp", "hist", "frac_bit_equal", "frac_within_1ulp", "nan_in_output")
                for k, t in p_out.items():
                    h = image_parity.ulp_histogram(t, kept["want"])
                    line["parity"][k] = {kk: h[kk] for kk in keys}
                # the intermediary of this arithmetic against FsrEasuF alone (F-strict, EXACT: 0 differing values), and — computed in THIS run,
                # not asserted in a note — the other arithmetics' final image on the same frame
                fsr.easu(p_src, p_mid, con=easu_con, flags=math_flags)
                h = image_parity.ulp_histogram(p_mid, kept["want_mid"])
                line["parity"]["intermediary_vs_FsrEasuF"] = {kk: h[kk] for kk in keys}
                for other, fl in (("exact", fsr.FLAG_MATH_EXACT), ("strict", fsr.FLAG_MATH_STRICT), ("f", 0)):
                    if other == args.math:
                        continue
                    fsr.easu(p_src, p_mid, con=easu_con, flags=fl)
                    fsr.rcas(p_mid, p_out["two_dispatch"], con=rcas_con, flags=fl)
                    h = image_parity.ulp_histogram(p_out["two_dispatch"], kept["want"])
                    line["parity"][{"exact": "exact_arithmetic", "strict": "strict_arithmetic", "f": "default_arithmetic"}[other] + "_two_dispatch"] = {kk: h[kk] for kk in keys}
        print(json.dumps(line), flush=True)
    if pipe is not None:
        pipe.close()
    if grouped:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
