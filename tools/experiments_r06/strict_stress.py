"""Round 6: the evidence F-strict's threshold rests on, at scale.  For `--seconds` of GPU time, frames of many kinds (generated on the device) at many
ratios:
  (1) easu(STRICT) vs easu(EXACT), RGBA16F and RGBA8 storage: differing stored values (must be 0), and how many values were compared;
  (2) on RGBA32F in / out: max |default - EXACT| / (2^-24 M), M = the largest |R|,|G|,|B| among the pixel's 12 taps — the measured bound the
      threshold kEasuStrictK = 32 is a multiple of — and its histogram in octaves.
Writes gpurun_out/r06_strict_stress.json.
"""
import argparse
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
import image_parity  # noqa: E402

fsr.load()
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(20260930)
NAT = torch.from_numpy(image_parity.natural_frame().astype(np.float32)).to(dev)


def make(kind, w, h):
    r = lambda *s: torch.rand(*s, device=dev, generator=g)
    if kind == "uniform":
        img = r(h, w, 4)
    elif kind == "smooth":
        lo = r(h // 8 + 2, w // 8 + 2, 4).permute(2, 0, 1)[None]
        img = torch.nn.functional.interpolate(lo, size=(h, w), mode="bilinear", align_corners=False)[0].permute(1, 2, 0) * 0.95 + r(h, w, 4) * 0.02
    elif kind == "blocks":
        lo = r(h // 16 + 2, w // 16 + 2, 4).permute(2, 0, 1)[None]
        img = torch.nn.functional.interpolate(lo, size=(h, w), mode="nearest")[0].permute(1, 2, 0)
    elif kind == "edges":
        y, x = torch.meshgrid(torch.arange(h, device=dev, dtype=torch.float32), torch.arange(w, device=dev, dtype=torch.float32), indexing="ij")
        a = r(3) * 3.1416
        f = 0.05 + r(3) * 0.4
        img = torch.stack([(torch.sin((x * torch.cos(a[c]) + y * torch.sin(a[c])) * f[c]) > 0).float() * (0.2 + 0.8 * r(1)) + 0.05 * r(1) for c in range(3)] + [torch.ones(h, w, device=dev)], -1)
        img = img + r(h, w, 4) * 0.01
    elif kind == "gradient":
        y, x = torch.meshgrid(torch.linspace(0, 1, h, device=dev), torch.linspace(0, 1, w, device=dev), indexing="ij")
        c = r(3, 3)
        img = torch.stack([(c[k, 0] * x + c[k, 1] * y + c[k, 2] * 0.3).clamp(0, 1) for k in range(3)] + [torch.ones(h, w, device=dev)], -1)
    elif kind == "dark":
        img = r(h, w, 4) ** 6
    elif kind == "hdr":
        img = torch.exp(torch.randn(h, w, 4, device=dev, generator=g) * 2.5).clamp(max=60000.0)
    elif kind == "text":  # sparse bright pixels / thin lines on dark, saturated colours
        img = (r(h, w, 1) > 0.93).float() * torch.tensor([1.0, 0.0, 0.5, 1.0], device=dev) + (r(h, 1, 1) > 0.9).float() * torch.tensor([0.0, 1.0, 0.0, 1.0], device=dev)
        img = img.clamp(0, 1).expand(h, w, 4).clone()
    else:  # natural: a random crop of the fixture, tinted and scaled, tiled to the size
        ny, nx = NAT.shape[:2]
        reps = (-(-h // ny), -(-w // nx), 1)
        img = NAT.repeat(*reps)[:h, :w] * (0.2 + 1.2 * r(1, 1, 4))
        img = torch.roll(img, shifts=(int(r(1) * ny), int(r(1) * nx)), dims=(0, 1))
    img = img.contiguous().clone()
    img[..., 3] = 1.0
    return img.half().contiguous()


def window_max(src32, ow, oh, con):
    ih, iw, _ = src32.shape
    c = np.asarray(con, np.uint32).view(np.float32)
    ox = torch.arange(ow, device=dev, dtype=torch.float32)
    oy = torch.arange(oh, device=dev, dtype=torch.float32)
    fx = torch.floor(ox * float(c[0]) + float(c[2])).to(torch.int64)
    fy = torch.floor(oy * float(c[1]) + float(c[3])).to(torch.int64)
    mag = src32[..., :3].abs().amax(dim=-1)
    M = torch.zeros(oh, ow, device=dev)
    for dy, dxs in ((-1, (0, 1)), (0, (-1, 0, 1, 2)), (1, (-1, 0, 1, 2)), (2, (0, 1))):
        yy = (fy + dy).clamp(0, ih - 1)
        for dx in dxs:
            xx = (fx + dx).clamp(0, iw - 1)
            M = torch.maximum(M, mag[yy][:, xx])
    return M


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=150.0)
    ap.add_argument("--seed", type=int, default=20260930, help="content generator seed (the committed logs: the default; *_seed2: 7)")
    args = ap.parse_args()
    g.manual_seed(args.seed)
    kinds = ["uniform", "smooth", "blocks", "edges", "gradient", "dark", "hdr", "text", "natural", "natural", "natural"]
    sizes = [(960, 540), (1280, 720), (1477, 831), (1001, 563), (640, 360)]
    ratios = [2.0, 2.0, 1.5, 1.3, 1.7, 1.0 + 1.0 / 3.0, 1.25, 1.9, 3.0]
    stats = {"frames": 0, "values_f16": 0, "values_u8": 0, "strict_vs_exact_differing_f16": 0, "strict_vs_exact_differing_u8": 0, "values_bound": 0,
             "max_delta_over_ulp_of_window_max": 0.0, "hist_octaves": [0] * 8, "per_kind_max": {}, "worst": None}
    edges = torch.tensor([1, 2, 4, 8, 16, 32, 64], device=dev, dtype=torch.float64)
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < args.seconds:
        kind = kinds[n % len(kinds)]
        iw, ih = sizes[(n // len(kinds)) % len(sizes)]
        ratio = ratios[(n // 3) % len(ratios)]
        ow, oh = (2 * iw, 2 * ih) if ratio == 2.0 else (int(iw * ratio), int(ih * ratio))
        img = make(kind, iw, ih)
        con = fsr.FsrEasuCon(iw, ih, iw, ih, ow, oh)
        # (1) strict vs exact, fp16 and rgba8
        ex, st = (torch.empty(oh, ow, 4, dtype=torch.float16, device=dev) for _ in range(2))
        fsr.easu(img, ex, con=con, flags=fsr.FLAG_MATH_EXACT)
        fsr.easu(img, st, con=con, flags=fsr.FLAG_MATH_STRICT | (fsr.FLAG_FRAMES_OVERLAP if n % 2 else 0))
        exi, sti = ex.view(torch.int16), st.view(torch.int16)
        bad = int(((exi != sti) & ~(torch.isnan(ex) & torch.isnan(st))).sum())
        stats["strict_vs_exact_differing_f16"] += bad
        stats["values_f16"] += ex.numel()
        if kind not in ("hdr",):
            u8 = (img.float().clamp(0, 1) * 255.0 + 0.5).floor().to(torch.uint8)
            ex8, st8 = (torch.empty(oh, ow, 4, dtype=torch.uint8, device=dev) for _ in range(2))
            fsr.easu(u8, ex8, con=con, flags=fsr.FLAG_MATH_EXACT)
            fsr.easu(u8, st8, con=con, flags=fsr.FLAG_MATH_STRICT)
            stats["strict_vs_exact_differing_u8"] += int((ex8 != st8).sum())
            stats["values_u8"] += ex8.numel()
        # (2) the bound, every third frame (it costs a dozen gathers)
        if n % 3 == 0:
            s32 = img.float()
            d, e = (torch.empty(oh, ow, 4, dtype=torch.float32, device=dev) for _ in range(2))
            fsr.easu(s32, d, con=con)
            fsr.easu(s32, e, con=con, flags=fsr.FLAG_MATH_EXACT)
            dv, ev = d[..., :3].double(), e[..., :3].double()
            fin = torch.isfinite(dv) & torch.isfinite(ev)
            delta = torch.where(fin, (dv - ev).abs(), torch.zeros_like(dv))
            M = window_max(s32, ow, oh, con).double().clamp_min(2.0 ** -126)[..., None]
            rr = delta / (M * 2.0 ** -24)
            mx = float(rr.max())
            stats["values_bound"] += rr.numel()
            stats["per_kind_max"][kind] = max(stats["per_kind_max"].get(kind, 0.0), round(mx, 3))
            if mx > stats["max_delta_over_ulp_of_window_max"]:
                stats["max_delta_over_ulp_of_window_max"] = round(mx, 3)
                stats["worst"] = {"kind": kind, "in": [iw, ih], "out": [ow, oh], "frame": n}
            b = torch.bucketize(rr.flatten(), edges, right=True)
            cnt = torch.bincount(b, minlength=8)
            for k in range(8):
                stats["hist_octaves"][k] += int(cnt[k])
        n += 1
        stats["frames"] = n
        if n % 50 == 0:
            print(json.dumps({k: v for k, v in stats.items() if k != "per_kind_max"}), flush=True)
    stats["hist_octave_edges"] = "<1, <2, <4, <8, <16, <32, <64, >=64"
    stats["seconds"] = round(time.perf_counter() - t0, 1)
    print(json.dumps(stats), flush=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "r06_strict_stress.json"), "w") as f:
        json.dump(stats, f, indent=1)
