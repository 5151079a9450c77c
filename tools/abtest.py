#!/usr/bin/env python3
"""A/B timing of library variants on the GPU box (tuning tool, not the bench contract).

  python tools/abtest.py --libs variants/a.so,variants/b.so --workloads 1080p_to_4k,1440p_to_4k --reps 3

Each variant runs in its own process (FSR1_HIP_LIB selects the library); per workload it prints one line with the
HIP-event time per launch (C-ABI stopwatch on the launch stream) of EASU, RCAS on the reused intermediary (`rcas`) and on an
image read from HBM (`rcas_cold`), the two-dispatch pair, and the fused launch — bench.py's buffer layout.  Variants are interleaved `reps` times so that box-to-box / thermal drift shows as spread
between repetitions rather than as a difference between variants.
"""
import argparse
import importlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(args):
    import numpy as np
    import torch
    import bench
    fsr = importlib.import_module("fidelityfx-fsr_amd")
    lib = fsr.load()
    # `lib%FSR1_FUSED_S2_STEPS=n` / `%FSR1_FUSED_S2_TALL=m` / `%FSR1_EASU_TALL=m` in --libs: the variant runs in the TEST library
    # (lib*_test.so, include/fsr1_hip_test.h) with its launch-shape hooks set; the product library has no such switches and reads no
    # tuning value from the environment
    if any(os.environ.get(k) for k in ("FSR1_FUSED_S2_STEPS", "FSR1_FUSED_S2_TALL", "FSR1_EASU_TALL")):
        lib = fsr._lib.load_test()
        fsr._lib._lib = lib  # every call of the package goes through the test library from here on
        if os.environ.get("FSR1_FUSED_S2_STEPS"):
            lib.fsr1_debug_fused_run_steps(int(os.environ["FSR1_FUSED_S2_STEPS"]))
        if os.environ.get("FSR1_FUSED_S2_TALL"):
            lib.fsr1_debug_fused_tall_tiles(int(os.environ["FSR1_FUSED_S2_TALL"]))
        if os.environ.get("FSR1_EASU_TALL"):
            lib.fsr1_debug_easu_tall_tiles(int(os.environ["FSR1_EASU_TALL"]))
    dev = torch.device("cuda", 0)
    flags = {"f": 0, "exact": fsr.FLAG_MATH_EXACT, "h": fsr.FLAG_MATH_PACKED_FP16}[args.math]
    if args.no_fast_paths:
        flags |= fsr.FLAG_NO_FAST_PATHS
    flags |= int(os.environ.get("FSR1_AB_FLAGS", "0"), 0)  # `lib@flags` in --libs: extra dispatch flag bits of this variant
    timer = fsr.Timer()
    for wl in args.workloads.split(","):
        in_w, in_h, out_w, out_h, frames = bench.WORKLOADS[wl]
        # bench.py's layout: inputs / outputs rotate over > 1 GiB, the intermediary is one reused buffer
        ring = max(2, -(-(1 << 30) // ((in_w * in_h + out_w * out_h) * 8 * frames)))
        base = torch.from_numpy(fsr.frames.synthetic_frame(in_w, in_h, k=1)).to(dev)
        srcs = [torch.stack([torch.roll(base, shifts=(3 * s + f, 5 * s + 2 * f), dims=(0, 1)) for f in range(frames)]).contiguous() for s in range(ring)]
        mid = torch.empty(frames, out_h, out_w, 4, dtype=torch.float16, device=dev)
        dsts = [torch.empty(frames, out_h, out_w, 4, dtype=torch.float16, device=dev) for _ in range(ring)]
        econ = fsr.FsrEasuCon(in_w, in_h, in_w, in_h, out_w, out_h)
        rcon = fsr.FsrRcasCon(0.25)
        fsr.easu(srcs[0], mid, con=econ, flags=flags)
        torch.cuda.synchronize()

        def easu(i): fsr.easu(srcs[i % ring], mid, con=econ, flags=flags)
        def rcas(i): fsr.rcas(mid, dsts[i % ring], con=rcon, flags=flags)
        def rcas_cold(i): fsr.rcas(dsts[(i + ring // 2) % ring], dsts[i % ring], con=rcon, flags=flags)  # an image read from HBM
        def pair(i): easu(i); rcas(i)
        def fused(i): fsr.easu_rcas_fused(srcs[i % ring], dsts[i % ring], easu_con=econ, rcas_con=rcon, flags=flags)

        def ms(fn, n):
            import time
            t0 = time.perf_counter()
            i = 0
            while time.perf_counter() - t0 < args.ramp:  # clock ramp + steady power state
                fn(i); i += 1
                if i % 64 == 0:
                    torch.cuda.synchronize()
            torch.cuda.synchronize()
            timer.start()
            for i in range(n):
                fn(i)
            timer.stop()
            return timer.elapsed_ms() / n

        n = max(10, int(args.launches / max(1, frames)))
        row = {"lib": os.path.basename(os.environ.get("FSR1_HIP_LIB", "default")) + ("@" + os.environ["FSR1_AB_FLAGS"] if os.environ.get("FSR1_AB_FLAGS") else "") + os.environ.get("FSR1_AB_TAG", ""),
               "workload": wl, "math": args.math}
        for name, fn in (("easu", easu), ("rcas", rcas), ("rcas_cold", rcas_cold), ("pair", pair), ("fused", fused)):
            if name in args.kernels.split(","):
                row[name + "_us"] = round(ms(fn, n) * 1e3, 2)
        print(json.dumps(row), flush=True)
        del srcs, mid, dsts
        torch.cuda.empty_cache()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--libs", default="")
    ap.add_argument("--workloads", default="1080p_to_4k")
    ap.add_argument("--kernels", default="easu,rcas,pair,fused")
    ap.add_argument("--math", default="f")
    ap.add_argument("--no-fast-paths", action="store_true")
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--launches", type=int, default=400)
    ap.add_argument("--ramp", type=float, default=0.25)
    ap.add_argument("--child", action="store_true")
    args = ap.parse_args()
    if args.child:
        return child(args)
    libs = [l for l in args.libs.split(",") if l] or [""]  # "@0x800" alone = the tree's library with those flag bits
    for rep in range(args.reps):
        for lib in libs:
            env = dict(os.environ)
            env.pop("FSR1_AB_FLAGS", None)
            env.pop("FSR1_AB_TAG", None)
            while "%" in lib:  # `path%NAME=VALUE`: the variant runs with that environment variable (e.g. %FSR1_FUSED_S2_STEPS=5)
                lib, kv = lib.rsplit("%", 1)
                k, v = kv.split("=", 1)
                env[k] = v
                env["FSR1_AB_TAG"] = env.get("FSR1_AB_TAG", "") + "%" + kv
            if "@" in lib:  # `path@flags`: the library (empty = the tree's) with extra dispatch flag bits, e.g. @0x10 = FSR1_FLAG_MATH_EXACT
                lib, env["FSR1_AB_FLAGS"] = lib.split("@", 1)
            if lib:
                env["FSR1_HIP_LIB"] = os.path.join(ROOT, lib)
            cmd = [sys.executable, os.path.abspath(__file__), "--child", "--workloads", args.workloads, "--kernels", args.kernels,
                   "--math", args.math, "--launches", str(args.launches), "--ramp", str(args.ramp)] + (["--no-fast-paths"] if args.no_fast_paths else [])
            r = subprocess.run(cmd, env=env, capture_output=True, text=True)
            for line in r.stdout.splitlines():
                if line.startswith("{"):
                    print(line, flush=True)
            if r.returncode:
                print(json.dumps({"lib": lib, "error": r.stderr[-400:]}), flush=True)


if __name__ == "__main__":
    main()
