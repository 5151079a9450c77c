#!/usr/bin/env python3
"""Locate the default-arithmetic EASU outliers (> 1 binary16 ULP) on frames.adversarial_frame and print their neighbourhood."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch, cpu_oracle
fsr = importlib.import_module("fidelityfx-fsr_amd"); fsr.load()
frames = importlib.import_module("fidelityfx-fsr_amd.frames")
port = cpu_oracle.port()
np.set_printoptions(linewidth=200, precision=6)
for (iw, ih, ow, oh) in ((96, 54, 192, 108), (192, 108, 384, 216), (80, 45, 120, 68)):
    for k in range(6):
        img = frames.adversarial_frame(iw, ih, k=k, dtype=np.float32)
        con = port.FsrEasuCon(iw, ih, iw, ih, ow, oh)
        want = port.easu_f(img, ow, oh, con)
        src = torch.from_numpy(img).cuda().half()
        out = torch.zeros(oh, ow, 4, dtype=torch.float16, device="cuda")
        fsr.easu(src, out, con=con)
        got = out.cpu().numpy().astype(np.float32)
        ex = torch.zeros_like(out); fsr.easu(src, ex, con=con, flags=fsr.FLAG_MATH_EXACT)
        d = cpu_oracle.half_ulp_diff(got, want.astype(np.float16).astype(np.float32))
        d = np.where(np.isfinite(want.astype(np.float16).astype(np.float32)), d, 0)
        bad = np.argwhere(d > 1)
        print("shape", (iw, ih, ow, oh), "k", k, "hist", np.bincount(np.minimum(d.ravel(), 8)).tolist(), "outliers", len(bad))
        for (y, x, c) in bad[:2]:
            fx = int(np.floor((x + 0.5) * iw / ow - 0.5)); fy = int(np.floor((y + 0.5) * ih / oh - 0.5))
            print("  pixel", (x, y, c), "got", got[y, x, :3], "want", want[y, x, :3], "exact-gpu", ex[y, x, :3].cpu().numpy())
            ys = slice(max(fy - 1, 0), fy + 3); xs = slice(max(fx - 1, 0), fx + 3)
            print("  channel", c, "window\n", img[ys, xs, c])
            lum = img[..., 0] * 0.5 + img[..., 2] * 0.5 + img[..., 1]
            print("  luma window\n", lum[ys, xs])
