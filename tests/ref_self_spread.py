#!/usr/bin/env python3
"""TEST INFRASTRUCTURE — how far do two conformant compilations of the REFERENCE lie from each other?

    python tests/ref_self_spread.py [--out profiles/r06_ref_self_spread.json]        (CPU only; needs /root/reference once, to build)

`oracle/_ref/libfsr1_ref.so` is the reference's headers compiled with -ffp-contract=off (the pinned parity target);
`oracle/_ref/libfsr1_ref_fma.so` is the same translation unit compiled with -ffp-contract=fast -mfma (oracle/build_ref.sh) — what a
shading-language compiler may do to the reference's `a*b+c` expressions unless they are marked `precise`.  Both are "the reference".
This script runs the chain

    FsrEasuF (ffx-fsr/ffx_fsr1.h:315-437) -> RTNE binary16 -> FsrRcasF (:684-769)

through both on the image-parity inputs (tests/test_gpu_image_parity.py: the synthetic generator and the natural-content fixture, RCAS
sharpness 0 / 0.25 / 1 stops) and publishes the binary16 ULP histograms of (a) the EASU intermediary and (b) the final image, twin
against pinned build.  It is the yardstick next to which README.md quotes the product's own distances: an integrator can see whether
"max N ULP" of an arithmetic is inside or outside what the reference does to itself under another legal compilation.
"""
import argparse
import importlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

CASES = [
    ("540p_to_1080p", "synthetic", (960, 540), (1920, 1080), 0.25),
    ("1080p_to_4k", "synthetic", (1920, 1080), (3840, 2160), 0.25),
    ("1080p_to_4k_sharp0", "synthetic", (1920, 1080), (3840, 2160), 0.0),
    ("1440p_to_4k", "synthetic", (2560, 1440), (3840, 2160), 0.25),
    ("natural_2x", "natural", (1477, 831), (2954, 1662), 0.25),
    ("natural_2x_sharp0", "natural", (1477, 831), (2954, 1662), 0.0),
    ("natural_2x_sharp1", "natural", (1477, 831), (2954, 1662), 1.0),
    ("natural_1p3x", "natural", (1477, 831), (1920, 1080), 0.25),
]


def histogram(a_f32, b_f32):
    import cpu_oracle
    d = cpu_oracle.half_ulp_diff(a_f32[..., :3], b_f32[..., :3])
    n = int(d.size)
    c0, c1, c2 = int((d == 0).sum()), int((d == 1).sum()), int((d == 2).sum())
    c4, cg = int(((d > 2) & (d <= 4)).sum()), int((d > 4).sum())
    return {"values": n, "max_ulp": int(d.max()), "hist": {"0": c0, "1": c1, "2": c2, "3-4": c4, ">4": cg},
            "frac_bit_equal": round(c0 / n, 6), "frac_within_1ulp": round((c0 + c1) / n, 6)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r06_ref_self_spread.json"))
    ap.add_argument("--cases", default="")
    args = ap.parse_args()
    import cpu_oracle
    import image_parity
    frames = importlib.import_module("fidelityfx-fsr_amd.frames")
    if not (cpu_oracle.have_ref() and cpu_oracle.have_ref_fma()):
        cpu_oracle.build(force=True)
    pinned, twin = cpu_oracle.ref(), cpu_oracle.ref_fma()
    report = {"what": "reference (-ffp-contract=fast -mfma) vs reference (-ffp-contract=off): binary16 ULP distance of the EASU "
                      "intermediary and of the final image of the chain FsrEasuF -> RTNE binary16 -> FsrRcasF; R, G, B only",
              "cases": {}}
    for name, content, (iw, ih), (ow, oh), sharp in CASES:
        if args.cases and name not in args.cases.split(","):
            continue
        if content == "natural":
            img = image_parity.natural_frame(iw, ih).astype(np.float32)
        else:
            img = frames.synthetic_frame(iw, ih, k=1).astype(np.float32)
        out_p, mid_p = image_parity.reference_chain(pinned, img, ow, oh, sharp, return_mid=True)
        out_t, mid_t = image_parity.reference_chain(twin, img, ow, oh, sharp, return_mid=True)
        # RCAS alone: the twin's RCAS on the PINNED build's intermediary (the stage-level spread)
        rcas_only = twin.rcas_f(mid_p, pinned.FsrRcasCon(sharp))
        report["cases"][name] = {"content": content, "in": [iw, ih], "out": [ow, oh], "sharpness_stops": sharp,
                                 "easu_intermediary": histogram(mid_t, mid_p), "rcas_on_identical_input": histogram(rcas_only, out_p),
                                 "final_image": histogram(out_t, out_p)}
        print(name, json.dumps(report["cases"][name]["final_image"]), flush=True)
    with open(args.out, "w") as f:
        json.dump(report, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
