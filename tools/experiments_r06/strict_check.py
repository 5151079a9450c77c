"""F-strict EASU vs EXACT, bit for bit, and the three arithmetics' kernel times (HIP events, 200 launches after a ramp)."""
import importlib
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
fsr = importlib.import_module("fidelityfx-fsr_amd")
frames = importlib.import_module("fidelityfx-fsr_amd.frames")
import image_parity  # noqa: E402

fsr.load()
dev = "cuda"


def time_kernel(fn, n=200, ramp=60):
    for _ in range(ramp):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n):
            fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1000.0 / n)
    return round(float(np.median(ts)), 2)


def check(name, img16, ow, oh, timing=True, overlap=False):
    ih, iw, _ = img16.shape
    res = {"shape": "%dx%d -> %dx%d" % (iw, ih, ow, oh)}
    con = fsr.FsrEasuCon(iw, ih, iw, ih, ow, oh)
    extra = fsr.FLAG_FRAMES_OVERLAP if overlap else 0
    for fmt in ("f16", "u8"):
        if fmt == "f16":
            src = torch.from_numpy(img16).to(dev)
            mk = lambda: torch.zeros(oh, ow, 4, dtype=torch.float16, device=dev)
            view = lambda t: t.view(torch.int16)
        else:
            src = torch.from_numpy(np.clip(np.rint(img16.astype(np.float32) * 255.0), 0, 255).astype(np.uint8)).to(dev)
            mk = lambda: torch.zeros(oh, ow, 4, dtype=torch.uint8, device=dev)
            view = lambda t: t
        ex, st, df = mk(), mk(), mk()
        fsr.easu(src, ex, con=con, flags=fsr.FLAG_MATH_EXACT | extra)
        fsr.easu(src, st, con=con, flags=fsr.FLAG_MATH_STRICT | extra)
        fsr.easu(src, df, con=con, flags=extra)
        torch.cuda.synchronize()
        nd = int((view(ex) != view(st)).sum())
        ndd = int((view(ex) != view(df)).sum())
        res[fmt] = {"strict_vs_exact_differing_values": nd, "default_vs_exact_differing_values": ndd, "values": ex.numel()}
        if nd:
            idx = torch.nonzero(view(ex) != view(st))[:5].tolist()
            res[fmt]["first_diffs"] = [(i, float(ex[i[0], i[1], i[2]]), float(st[i[0], i[1], i[2]])) for i in idx]
        if timing and fmt == "f16":
            res["us"] = {"default": time_kernel(lambda: fsr.easu(src, df, con=con, flags=extra)),
                         "strict": time_kernel(lambda: fsr.easu(src, st, con=con, flags=fsr.FLAG_MATH_STRICT | extra)),
                         "exact": time_kernel(lambda: fsr.easu(src, ex, con=con, flags=fsr.FLAG_MATH_EXACT | extra))}
            res["us"]["strict_over_default"] = round(res["us"]["strict"] / res["us"]["default"], 3)
    print(name, json.dumps(res), flush=True)
    return res


if __name__ == "__main__":
    out = {}
    nat = image_parity.natural_frame()
    for name, (iw, ih, ow, oh) in image_parity.SHAPES.items():
        out["synthetic_" + name] = check("synthetic_" + name, frames.synthetic_frame(iw, ih, k=7), ow, oh)
    out["synthetic_1080p_to_4k_overlap"] = check("synthetic_1080p_to_4k_overlap", frames.synthetic_frame(1920, 1080, k=7), 3840, 2160, overlap=True)
    out["natural_x2"] = check("natural_x2", nat, 2954, 1662)
    out["natural_x2_overlap"] = check("natural_x2_overlap", nat, 2954, 1662, overlap=True)
    out["natural_1p3"] = check("natural_1p3", nat, 1920, 1080)
    out["natural_x1p5"] = check("natural_x1p5", nat, 2216, 1247)
    out["natural_tile4k_x2"] = check("natural_tile4k_x2", np.tile(nat, (2, 2, 1))[:1080, :1920].copy(), 3840, 2160)
    out["natural_tile4k_x1p5"] = check("natural_tile4k_x1p5", np.tile(nat, (2, 2, 1))[:1440, :2560].copy(), 3840, 2160)
    out["adversarial_x2"] = check("adversarial_x2", frames.adversarial_frame(960, 540, k=1), 1920, 1080, timing=False)
    out["adversarial_x1p5"] = check("adversarial_x1p5", frames.adversarial_frame(960, 540, k=2), 1440, 810, timing=False)
    out["ragged"] = check("ragged", frames.synthetic_frame(333, 211, k=2), 666, 422, timing=False)
    out["ragged_1p7"] = check("ragged_1p7", frames.synthetic_frame(333, 211, k=2), 567, 359, timing=False)
    out["tiny"] = check("tiny", frames.synthetic_frame(5, 3, k=2), 10, 6, timing=False)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "r06_strict_check.json"), "w") as f:
        json.dump(out, f, indent=1)
