"""Round 6, step 0 of "F-strict": how far is the default EASU arithmetic from FsrEasuF's operation order, in binary32?

Runs easu(default) and easu(EXACT) with RGBA32F in / out (the arithmetic after the load is the same as with RGBA16F storage)
on inputs holding binary16-representable values, and reports |default - EXACT| per value
  * in binary32 ULPs of the EXACT value,
  * relative to M = max |c| over the pixel's 12-tap window (x 2^-24: "ULPs of the window's magnitude"),
so that a threshold for the rounding-boundary test can be chosen.  GPU only; writes gpurun_out/r06_default_vs_exact.json.
"""
import importlib
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
fsr = importlib.import_module("fidelityfx-fsr_amd")
frames = importlib.import_module("fidelityfx-fsr_amd.frames")
import image_parity  # noqa: E402

fsr.build(); fsr.load()
dev = "cuda"


def window_max(src, ow, oh, con):
    """M[oy, ox] = max |c| over R,G,B of the 12 taps of output pixel (ox, oy) — positions per ffx_fsr1.h:324-342, clamp-to-edge."""
    ih, iw, _ = src.shape
    c = np.asarray(con, np.uint32).view(np.float32)
    ox = torch.arange(ow, device=dev, dtype=torch.float32)
    oy = torch.arange(oh, device=dev, dtype=torch.float32)
    fx = torch.floor(ox * float(c[0]) + float(c[2])).to(torch.int64)
    fy = torch.floor(oy * float(c[1]) + float(c[3])).to(torch.int64)
    mag = src[..., :3].abs().amax(dim=-1)  # (ih, iw)
    M = torch.zeros(oh, ow, device=dev)
    for dy, dxs in ((-1, (0, 1)), (0, (-1, 0, 1, 2)), (1, (-1, 0, 1, 2)), (2, (0, 1))):
        yy = (fy + dy).clamp(0, ih - 1)
        for dx in dxs:
            xx = (fx + dx).clamp(0, iw - 1)
            M = torch.maximum(M, mag[yy][:, xx])
    return M


def measure(name, img16, ow, oh):
    ih, iw, _ = img16.shape
    src = torch.from_numpy(img16.astype(np.float32)).to(dev)
    con = fsr.FsrEasuCon(iw, ih, iw, ih, ow, oh)
    d = torch.zeros(oh, ow, 4, dtype=torch.float32, device=dev)
    e = torch.zeros_like(d)
    fsr.easu(src, d, con=con)
    fsr.easu(src, e, con=con, flags=fsr.FLAG_MATH_EXACT)
    torch.cuda.synchronize()
    dv, ev = d[..., :3], e[..., :3]
    fin = torch.isfinite(dv) & torch.isfinite(ev)
    delta = (dv.double() - ev.double()).abs()
    delta = torch.where(fin, delta, torch.zeros_like(delta))
    # binary32 ULP of the EXACT value (normal range; below 2^-126 use the subnormal spacing)
    ex = torch.frexp(ev.abs().clamp_min(2.0 ** -126))[1].double() - 1.0
    ulp = torch.pow(torch.tensor(2.0, dtype=torch.float64, device=dev), ex - 23.0)
    r_ulp = delta / ulp
    M = window_max(src, ow, oh, con).double().clamp_min(2.0 ** -126)[..., None]
    r_mag = delta / (M * 2.0 ** -24)
    # binary16 rounding flips between the two (what F-strict has to catch)
    flips = int(((dv.half().view(torch.int16) != ev.half().view(torch.int16)) & fin).sum())
    # among the flips: the largest delta relative to M and to the value
    fl = (dv.half().view(torch.int16) != ev.half().view(torch.int16)) & fin
    q = lambda t, p: float(torch.quantile(t.flatten()[:: max(1, t.numel() // 4_000_000)].float(), p))
    res = {
        "shape": "%dx%d -> %dx%d" % (iw, ih, ow, oh), "values": int(delta.numel()), "binary16_flips": flips,
        "nonfinite": int((~fin).sum()),
        "delta_in_ulp32_of_value": {"mean": float(r_ulp.mean()), "p99": q(r_ulp, 0.99), "p9999": q(r_ulp, 0.9999), "max": float(r_ulp.max())},
        "delta_in_ulp32_of_window_max": {"mean": float(r_mag.mean()), "p99": q(r_mag, 0.99), "p9999": q(r_mag, 0.9999), "max": float(r_mag.max())},
        "at_flips": {"max_delta_ulp32_of_value": float(r_ulp[fl].max()) if flips else 0.0, "max_delta_ulp32_of_window_max": float(r_mag[fl].max()) if flips else 0.0},
    }
    # histogram of delta / (M 2^-24) in octaves
    edges = [0, 1, 2, 4, 8, 16, 32, 64, 128, 256, 1e30]
    res["hist_window_max_ulps"] = {("<%g" % edges[i + 1]): int(((r_mag >= edges[i]) & (r_mag < edges[i + 1])).sum()) for i in range(len(edges) - 1)}
    res["hist_value_ulps"] = {("<%g" % edges[i + 1]): int(((r_ulp >= edges[i]) & (r_ulp < edges[i + 1])).sum()) for i in range(len(edges) - 1)}
    # what a threshold e = K * M * 2^-24 would flag (fraction of PIXELS with any channel within e of a binary16 rounding boundary)
    for K in (16, 32, 64, 128, 256):
        eps = (K * 2.0 ** -24) * M
        lo = (ev.double() - eps).float().half().view(torch.int16)
        hi = (ev.double() + eps).float().half().view(torch.int16)
        res["flagged_pixels_K%d" % K] = round(float((lo != hi).any(dim=-1).float().mean()), 5)
    print(name, json.dumps(res), flush=True)
    return res


if __name__ == "__main__":
    out = {}
    for name, (iw, ih, ow, oh) in image_parity.SHAPES.items():
        if name == "4k_to_8k":
            continue  # 8K RGBA32F + float64 temporaries: skip (same kernel as 1080p -> 4K)
        out["synthetic_" + name] = measure("synthetic_" + name, frames.synthetic_frame(iw, ih, k=7), ow, oh)
    nat = image_parity.natural_frame()
    out["natural_831p_to_1080p"] = measure("natural_831p_to_1080p", nat, 1920, 1080)
    out["natural_831p_x2"] = measure("natural_831p_x2", nat, 2954, 1662)
    out["natural_831p_x1p5"] = measure("natural_831p_x1p5", nat, 2216, 1247)
    adv = frames.adversarial_frame(960, 540, k=1)
    out["adversarial_540p_x2"] = measure("adversarial_540p_x2", adv, 1920, 1080)
    out["adversarial_540p_x1p5"] = measure("adversarial_540p_x1p5", adv, 1440, 810)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "r06_default_vs_exact.json"), "w") as f:
        json.dump(out, f, indent=1)
