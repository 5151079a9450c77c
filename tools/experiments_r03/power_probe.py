#!/usr/bin/env python3
"""Package power and shader clock while one kernel loops (rocm-smi sampled from a side thread), per library variant.
  python tools/experiments_r03/power_probe.py --libs @0,variants/libfsr1_mfma_default.so --kernel easu --seconds 6
Prints one JSON line per variant: launches/s, median of the sampled power (W) and sclk (MHz) readings."""
import argparse, importlib, json, os, re, subprocess, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def child(args):
    import torch
    import bench
    fsr = importlib.import_module("fidelityfx-fsr_amd")
    fsr.load()
    in_w, in_h, out_w, out_h, frames = bench.WORKLOADS[args.workload]
    dev = torch.device("cuda", 0)
    base = torch.from_numpy(fsr.frames.synthetic_frame(in_w, in_h, k=1)).to(dev)
    ring = 8
    srcs = [torch.stack([torch.roll(base, shifts=(3 * s + f, 5 * s), dims=(0, 1)) for f in range(frames)]).contiguous() for s in range(ring)]
    mid = torch.empty(frames, out_h, out_w, 4, dtype=torch.float16, device=dev)
    dsts = [torch.empty_like(mid) for _ in range(ring)]
    econ, rcon = fsr.FsrEasuCon(in_w, in_h, in_w, in_h, out_w, out_h), fsr.FsrRcasCon(0.25)
    flags = int(os.environ.get("FSR1_AB_FLAGS", "0"), 0)
    fn = {"easu": lambda i: fsr.easu(srcs[i % ring], mid, con=econ, flags=flags),
          "rcas": lambda i: fsr.rcas(mid, dsts[i % ring], con=rcon, flags=flags),
          "fused": lambda i: fsr.easu_rcas_fused(srcs[i % ring], dsts[i % ring], easu_con=econ, rcas_con=rcon, flags=flags)}[args.kernel]
    samples, stop = [], threading.Event()

    def sample():
        while not stop.is_set():
            try:
                out = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True, timeout=10).stdout
                p = re.search(r"Power[^\n]*?:\s*([0-9.]+)", out)
                c = re.search(r"sclk clock level[^\n]*?\(([0-9.]+)Mhz\)", out)
                samples.append((float(p.group(1)) if p else None, float(c.group(1)) if c else None))
            except Exception as e:  # noqa: BLE001
                samples.append((None, None))
            time.sleep(0.3)
    fn(0); torch.cuda.synchronize()
    th = threading.Thread(target=sample); th.start()
    t0 = time.perf_counter(); n = 0
    while time.perf_counter() - t0 < args.seconds:
        for _ in range(64):
            fn(n); n += 1
        torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    stop.set(); th.join()
    mid_s = samples[len(samples) // 4:] or samples  # skip the ramp
    pw = sorted(s[0] for s in mid_s if s[0] is not None); ck = sorted(s[1] for s in mid_s if s[1] is not None)
    print(json.dumps({"lib": os.path.basename(os.environ.get("FSR1_HIP_LIB", "default")) + "@" + os.environ.get("FSR1_AB_FLAGS", "0"), "kernel": args.kernel,
                      "workload": args.workload, "us_per_launch_wall": round(dt / n * 1e6, 2), "samples": len(samples),
                      "power_W_median": pw[len(pw) // 2] if pw else None, "power_W_max": pw[-1] if pw else None,
                      "sclk_MHz_median": ck[len(ck) // 2] if ck else None, "sclk_MHz_min": ck[0] if ck else None}), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--libs", default="@0")
    ap.add_argument("--kernel", default="easu")
    ap.add_argument("--workload", default="1080p_to_4k")
    ap.add_argument("--seconds", type=float, default=6.0)
    ap.add_argument("--child", action="store_true")
    args = ap.parse_args()
    if args.child:
        return child(args)
    for lib in args.libs.split(","):
        env = dict(os.environ); env.pop("FSR1_AB_FLAGS", None)
        if "@" in lib:
            lib, env["FSR1_AB_FLAGS"] = lib.split("@", 1)
        if lib:
            env["FSR1_HIP_LIB"] = os.path.join(ROOT, lib)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", "--kernel", args.kernel, "--workload", args.workload, "--seconds", str(args.seconds)],
                           env=env, capture_output=True, text=True)
        print("\n".join(l for l in r.stdout.splitlines() if l.startswith("{")) or json.dumps({"lib": lib, "error": r.stderr[-300:]}), flush=True)


if __name__ == "__main__":
    main()
