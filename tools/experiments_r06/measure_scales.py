"""Round 6, F-strict step 0b: which scale bounds |default - EXACT| tightest?  Run with the `alpha_aw` variant
(tools/experiments_r06/alpha_is_aw.patch: the default arithmetic stores the window's weight sum aW as alpha):

    FSR1_HIP_LIB=variants/libfsr1_alpha_aw.so python tools/experiments_r06/measure_scales.py

Scales compared, all x 2^-24:  M = max |c| over the 12 taps and R,G,B;  M / aW;  M_c = per-channel max;  M_c / aW.
For each: max and p99.99 of delta / scale over a corpus, and the fraction of pixels a threshold of 2.5 x that max would flag.
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

fsr.load()
dev = "cuda"


def window_max(src, ow, oh, con):
    ih, iw, _ = src.shape
    c = np.asarray(con, np.uint32).view(np.float32)
    ox = torch.arange(ow, device=dev, dtype=torch.float32)
    oy = torch.arange(oh, device=dev, dtype=torch.float32)
    fx = torch.floor(ox * float(c[0]) + float(c[2])).to(torch.int64)
    fy = torch.floor(oy * float(c[1]) + float(c[3])).to(torch.int64)
    mag = src[..., :3].abs()
    M = torch.zeros(oh, ow, 3, device=dev)
    for dy, dxs in ((-1, (0, 1)), (0, (-1, 0, 1, 2)), (1, (-1, 0, 1, 2)), (2, (0, 1))):
        yy = (fy + dy).clamp(0, ih - 1)
        for dx in dxs:
            xx = (fx + dx).clamp(0, iw - 1)
            M = torch.maximum(M, mag[yy][:, xx])
    return M


ACC = {}


def measure(name, img16, ow, oh):
    ih, iw, _ = img16.shape
    src = torch.from_numpy(img16.astype(np.float32)).to(dev)
    con = fsr.FsrEasuCon(iw, ih, iw, ih, ow, oh)
    d = torch.zeros(oh, ow, 4, dtype=torch.float32, device=dev)
    e = torch.zeros_like(d)
    fsr.easu(src, d, con=con)
    fsr.easu(src, e, con=con, flags=fsr.FLAG_MATH_EXACT)
    torch.cuda.synchronize()
    dv, ev = d[..., :3].double(), e[..., :3].double()
    aW = d[..., 3:4].double()
    fin = torch.isfinite(dv) & torch.isfinite(ev)
    delta = torch.where(fin, (dv - ev).abs(), torch.zeros_like(dv))
    Mc = window_max(src, ow, oh, con).double().clamp_min(2.0 ** -126)
    M = Mc.amax(dim=-1, keepdim=True)
    u = 2.0 ** -24
    scales = {"M": M * u, "M/aW": M * u / aW.abs().clamp_min(1e-30), "Mc": Mc * u, "Mc/aW": Mc * u / aW.abs().clamp_min(1e-30)}
    res = {"shape": "%dx%d -> %dx%d" % (iw, ih, ow, oh), "aW": {"min": float(aW.min()), "p001": float(torch.quantile(aW.flatten()[::7].float(), 0.001)),
                                                             "median": float(aW.median()), "max": float(aW.max())}}
    for k, s in scales.items():
        r = delta / s
        res[k] = {"max": round(float(r.max()), 3), "p9999": round(float(torch.quantile(r.flatten()[:: max(1, r.numel() // 4_000_000)].float(), 0.9999)), 3)}
        ACC[k] = max(ACC.get(k, 0.0), res[k]["max"])
    # flagged fractions for thresholds at 2.5 x the corpus-wide max known so far from the first experiment (M: 12.9 -> 32); the others are
    # reported for a range of K so the table can be read once the maxima are known
    for k, s in scales.items():
        for K in (8, 12, 16, 24, 32, 48, 64):
            eps = K * s
            lo = (ev - eps).float().half().view(torch.int16)
            hi = (ev + eps).float().half().view(torch.int16)
            res.setdefault("flagged_" + k, {})[str(K)] = round(float((lo != hi).any(dim=-1).float().mean()), 5)
    print(name, json.dumps(res), flush=True)
    return res


def rnd_frame(w, h, seed, kind):
    g = np.random.default_rng(seed)
    if kind == "uniform":
        img = g.random((h, w, 4), dtype=np.float32)
    elif kind == "smooth":
        base = g.random((h // 8 + 2, w // 8 + 2, 4), dtype=np.float32)
        img = np.kron(base, np.ones((8, 8, 1), np.float32))[:h, :w] * 0.9 + g.random((h, w, 4), dtype=np.float32) * 0.02
    elif kind == "edges":
        y, x = np.mgrid[0:h, 0:w]
        ang = g.random(3) * 3.14
        img = np.stack([(np.sin((x * np.cos(a) + y * np.sin(a)) * f) > 0).astype(np.float32) * 0.9 + 0.05 for a, f in zip(ang, (0.21, 0.13, 0.37))] + [np.ones((h, w), np.float32)], -1)
        img += g.random((h, w, 4), dtype=np.float32) * 0.01
    elif kind == "dark":
        img = g.random((h, w, 4), dtype=np.float32) ** 6
    elif kind == "hdr":
        img = np.exp(g.normal(0, 3, (h, w, 4))).astype(np.float32)
        img = np.minimum(img, 60000.0)
    img[..., 3] = 1.0
    return img.astype(np.float16)


if __name__ == "__main__":
    out = {}
    nat = image_parity.natural_frame()
    out["natural_x2"] = measure("natural_x2", nat, 2954, 1662)
    out["natural_1p3"] = measure("natural_1p3", nat, 1920, 1080)
    out["natural_x1p5"] = measure("natural_x1p5", nat, 2216, 1247)
    out["synthetic_1080p_x2"] = measure("synthetic_1080p_x2", frames.synthetic_frame(1920, 1080, k=3), 3840, 2160)
    out["synthetic_1440p_x1p5"] = measure("synthetic_1440p_x1p5", frames.synthetic_frame(2560, 1440, k=3), 3840, 2160)
    for kind in ("uniform", "smooth", "edges", "dark", "hdr"):
        for seed in range(2):
            for (w, h, ow, oh) in ((1280, 720, 2560, 1440), (1280, 720, 1920, 1080), (1280, 720, 2176, 1224), (1281, 721, 1665, 937)):
                n = "%s_s%d_%dx%d" % (kind, seed, ow, oh)
                out[n] = measure(n, rnd_frame(w, h, seed * 100 + ow, kind), ow, oh)
    out["adversarial_x2"] = measure("adversarial_x2", frames.adversarial_frame(960, 540, k=1), 1920, 1080)
    out["adversarial_x1p5"] = measure("adversarial_x1p5", frames.adversarial_frame(960, 540, k=2), 1440, 810)
    out["corpus_max"] = ACC
    print("CORPUS MAX", json.dumps(ACC))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "r06_scales.json"), "w") as f:
        json.dump(out, f, indent=1)
