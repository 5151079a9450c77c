"""Regenerates tests/golden/*.npz and con_kat.json from oracle/_ref — the reference's own ffx_a.h /
ffx_fsr1.h compiled verbatim (oracle/build_ref.sh).  Needs /root/reference; run in the build
container:  python tests/golden/gen_golden.py

Each pixel fixture holds: the input frame (fp16-representable float32 RGBA), the FsrEasuCon words,
and the CPU-evaluated reference outputs
  easu_f / easu_h                  FsrEasuF / FsrEasuH of the input
  mid                              easu_f rounded to binary16 (the two-pass pipeline's intermediary)
  rcas_f[flags] / rcas_h[flags]    FsrRcasF / FsrRcasH of `mid` for flags in {0, DENOISE, ALPHA, ALPHA|DENOISE}
"""
import importlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import cpu_oracle  # noqa: E402

frames = importlib.import_module("fidelityfx-fsr_amd.frames")
HERE = os.path.dirname(os.path.abspath(__file__))

# name, (in_w, in_h), (out_w, out_h), frame index, sharpness
CASES = [
    ("perf_2p0x", (48, 27), (96, 54), 0, 0.25),
    ("balanced_1p7x", (40, 23), (68, 39), 1, 0.2),
    ("quality_1p5x", (43, 29), (64, 43), 2, 0.0),
    ("ultra_1p3x", (50, 31), (65, 40), 3, 1.0),
    ("odd_2p0x_ragged", (37, 23), (74, 46), 4, 0.25),
]

EASU_SHAPES = [
    (960, 540, 1920, 1080), (1477, 831, 1920, 1080), (1920, 1080, 3840, 2160), (2560, 1440, 3840, 2160),
    (2259, 1270, 3840, 2160), (2954, 1662, 3840, 2160), (3840, 2160, 7680, 4320), (5120, 2880, 7680, 4320),
    (64, 36, 128, 72), (1, 1, 1, 1), (7, 5, 3, 2),
]
RCAS_STOPS = [0.0, 0.2, 0.25, 0.5, 1.0, 2.0, 0.1, 1.37]
HALF_INPUTS = [1.0, 0.870550573, 65504.0, 65520.0, 1e-8, 6e-8, -0.3333, float("inf"), float("nan"), 0.0, -0.0,
               6.103515625e-05, 6.0e-05, 5.9604645e-08, 2.9e-08, 1e30, -1e30, 0.5, 0.25, 3.0e-05]


# Colour stages (ffx_fsr1.h:986-1199): (stage bits, image key) -> CPU-evaluated reference output.
COLOR_CASES = [(1, "hdr"), (2, "ldr"), (4, "ldr"), (8, "ldr"), (16, "ldr"), (8 | 32, "ldr"), (16 | 32, "ldr"), (2 | 8, "ldr"),
               (2 | 4, "ldr"), (1 | 2 | 4, "hdr"), (2 | 16 | 32, "ldr"), (1 | 2 | 8, "hdr")]
COLOR_PARAMS = dict(amount=0.75, bias=0.0, frame=3, noise_offset=(5, -3))


def color_inputs():
    """Deterministic inputs of the colour-stage fixture: an LDR frame with exact 0 / 1 / grid values, an HDR frame,
    and a 2-slice 8x8 signed noise tile (rgb = grain in [-0.5, 0.5], a = dither in [0, 1))."""
    ldr = frames.synthetic_frame(40, 24, k=6, dtype=np.float32)
    ldr[0, :8, :3] = np.array([0.0, 1.0, 0.5, 1.0 / 255.0, (7.0 / 255.0) ** 2, 0.999, 1e-7, 0.25], np.float32)[:, None]
    ldr = ldr.astype(np.float16).astype(np.float32)
    hdr = (frames.synthetic_frame(40, 24, k=9, dtype=np.float32).astype(np.float64) ** 3 * 4000.0).astype(np.float16).astype(np.float32)
    hdr[..., 3] = 1.0
    s = np.uint32(12345)
    vals = np.empty(2 * 8 * 8 * 4, np.float32)
    for i in range(vals.size):
        s = np.uint32((int(s) * 1664525 + 1013904223) & 0xFFFFFFFF)
        vals[i] = (int(s) >> 8) / float(1 << 24)
    noise = vals.reshape(2, 8, 8, 4)
    noise[..., :3] -= 0.5
    noise = noise.astype(np.float16).astype(np.float32)
    return ldr, hdr, noise


def write_color(R):
    ldr, hdr, noise = color_inputs()
    out = {"ldr": ldr.astype(np.float16), "hdr": hdr.astype(np.float16), "noise": noise.astype(np.float16)}
    for st, key in COLOR_CASES:
        out["out_%d_%s" % (st, key)] = R.color_f({"ldr": ldr, "hdr": hdr}[key], st, noise=noise, **COLOR_PARAMS)
        # the half-precision entry points (FsrSrtmH, FsrLfgaH, FsrTepdC8H ...) on the same inputs
        out["outh_%d_%s" % (st, key)] = R.color_h({"ldr": ldr, "hdr": hdr}[key], st, noise=noise, **COLOR_PARAMS).astype(np.float16)
    out["dit"] = np.array([[R.FsrTepdDitF(x, y, f) for x in (0, 1, 17, 3839, 7679)] for y, f in ((0, 0), (5, 1), (2159, 7), (4319, 1000))],
                          np.float32)
    np.savez_compressed(os.path.join(HERE, "color_stages.npz"), **out)
    print("color_stages written")


def main():
    R = cpu_oracle.ref()
    kat = {"easu": [], "easu_offset": [], "rcas": [], "half": []}
    for iw, ih, ow, oh in EASU_SHAPES:
        kat["easu"].append({"args": [iw, ih, iw, ih, ow, oh], "con": ["%08x" % v for v in R.FsrEasuCon(iw, ih, iw, ih, ow, oh)]})
    for args in [(1280, 720, 1920, 1080, 2560, 1440, 16, 8), (1600, 900, 1920, 1080, 3840, 2160, 0, 0),
                 (900, 500, 1024, 600, 1800, 1000, 3.5, 7.25)]:
        kat["easu_offset"].append({"args": list(args), "con": ["%08x" % v for v in R.FsrEasuConOffset(*args)]})
    for s in RCAS_STOPS:
        kat["rcas"].append({"stops": s, "con": ["%08x" % v for v in R.FsrRcasCon(s)]})
    for f in HALF_INPUTS:
        kat["half"].append({"f_bits": "%08x" % np.float32(f).view(np.uint32), "h": "%04x" % R.AU1_AH1_AF1(f)})
    with open(os.path.join(HERE, "con_kat.json"), "w") as fh:
        json.dump(kat, fh, indent=1)

    for name, (iw, ih), (ow, oh), k, stops in CASES:
        img = frames.synthetic_frame(iw, ih, k=k, dtype=np.float32)
        img[..., 3] = frames.synthetic_frame(iw, ih, k=k + 7, dtype=np.float32)[..., 1]  # non-trivial alpha
        con = R.FsrEasuCon(iw, ih, iw, ih, ow, oh)
        rcon = R.FsrRcasCon(stops)
        out = {"input": img.astype(np.float16), "con": con, "rcas_con": rcon, "stops": np.float32(stops)}
        ef = R.easu_f(img, ow, oh, con)
        out["easu_f"] = ef
        out["easu_h"] = R.easu_h(img, ow, oh, con).astype(np.float16)
        mid = ef.astype(np.float16)
        mid[..., 3] = frames.synthetic_frame(ow, oh, k=k + 11, dtype=np.float16)[..., 2]  # alpha to pass through
        out["mid"] = mid
        m32 = mid.astype(np.float32)
        for fl in (0, 1, 2, 3):
            out["rcas_f_%d" % fl] = R.rcas_f(m32, rcon, fl)
            out["rcas_h_%d" % fl] = R.rcas_h(m32, rcon, fl).astype(np.float16)
        out["rcas_f_hdr"] = R.rcas_f(m32, rcon, 4)
        out["easu_f_hdr"] = R.easu_f(img, ow, oh, con, 4)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
        print(name, "written")

    write_color(R)

    # SURVEY.md Appendix B.2 KAT frame, fp32 in/out, intermediate not rounded
    img = frames.kat_frame_64x36()
    con = R.FsrEasuCon(64, 36, 64, 36, 128, 72)
    e = R.easu_f(img, 128, 72, con)
    r = R.rcas_f(e, R.FsrRcasCon(0.25))
    np.savez_compressed(os.path.join(HERE, "kat_b2.npz"), input=img, con=con, easu_f=e, rcas_f=r)
    print("kat_b2 written; rcas sum =", r.astype(np.float64).sum())


if __name__ == "__main__":
    main()
