"""Round 6: how many pixels per TILE fail F-strict's rounding-boundary test — the re-evaluation costs one pass of ~520 wave-instructions per
started group of 64 queued pixels of a tile, so the distribution per tile, not the fraction per image, is what its time follows.
The test is re-stated in torch on the default arithmetic's binary32 output (RGBA32F out: the value before the store conversion):
e = K * 2^-24 * M (K = 48 until the adversarial search, 56 since) capped per channel by the dering interval's width, flagged when binary16(x - e) != binary16(x + e) in some channel.
Writes gpurun_out/r06_strict_flag_census.json."""
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
import image_parity  # noqa: E402

dev = "cuda"


def census(src16, ow, oh, K, tile_w, tile_h, shift):
    ih, iw, _ = src16.shape
    s32 = src16.float().contiguous()
    con = fsr.FsrEasuCon(iw, ih, iw, ih, ow, oh)
    out = torch.empty(oh, ow, 4, dtype=torch.float32, device=dev)
    fsr.easu(s32, out, con=con)
    c = np.asarray(con, np.uint32).view(np.float32)
    fx = torch.floor(torch.arange(ow, device=dev, dtype=torch.float32) * float(c[0]) + float(c[2])).to(torch.int64)
    fy = torch.floor(torch.arange(oh, device=dev, dtype=torch.float32) * float(c[1]) + float(c[3])).to(torch.int64)
    gx = [(fx + d).clamp(0, iw - 1) for d in (-1, 0, 1, 2)]
    gy = [(fy + d).clamp(0, ih - 1) for d in (-1, 0, 1, 2)]
    mag = s32[..., :3].abs().amax(dim=-1)
    M = torch.zeros(oh, ow, device=dev)
    for iy, dxs in ((0, (1, 2)), (1, (0, 1, 2, 3)), (2, (0, 1, 2, 3)), (3, (1, 2))):
        rows = mag[gy[iy]]
        for ix in dxs:
            M = torch.maximum(M, rows[:, gx[ix]])
    rgb = s32[..., :3]
    blk = torch.stack([rgb[gy[1]][:, gx[1]], rgb[gy[1]][:, gx[2]], rgb[gy[2]][:, gx[1]], rgb[gy[2]][:, gx[2]]])
    width = blk.amax(dim=0) - blk.amin(dim=0)
    e = torch.minimum((K * 2.0 ** -24 * M)[..., None].expand(-1, -1, 3), width)
    x = out[..., :3]
    flagged = ((x - e).half().view(torch.int16) != (x + e).half().view(torch.int16)).any(dim=-1)
    ty = torch.div(torch.arange(oh, device=dev) + shift, tile_h, rounding_mode="floor")
    tx = torch.div(torch.arange(ow, device=dev) + shift, tile_w, rounding_mode="floor")
    ntx, nty = int(tx.max()) + 1, int(ty.max()) + 1
    per_tile = torch.zeros(ntx * nty, device=dev).scatter_add(0, (ty[:, None] * ntx + tx[None, :]).flatten(), flagged.flatten().float())
    n = per_tile.cpu().numpy()
    passes = np.ceil(n / 64.0)
    return {"K": K, "tile": [tile_w, tile_h], "tiles": int(n.size), "flagged_fraction": round(float(flagged.float().mean()), 5),
            "per_tile_mean": round(float(n.mean()), 1), "per_tile_p10_p50_p90_max": [float(np.percentile(n, q)) for q in (10, 50, 90, 100)],
            "tiles_with_0": round(float((n == 0).mean()), 4), "tiles_over_64": round(float((n > 64).mean()), 4), "tiles_over_128": round(float((n > 128).mean()), 4),
            "wave_passes_per_tile": round(float(passes.mean()), 3), "dense_wave_passes_per_tile": round(float(n.sum() / 64.0 / n.size), 3)}


if __name__ == "__main__":
    fsr.load()
    doc = {}
    synth = torch.from_numpy(fsr.frames.synthetic_frame(1920, 1080, k=1)).to(dev)
    synth1440 = torch.from_numpy(fsr.frames.synthetic_frame(2560, 1440, k=1)).to(dev)
    nat = torch.from_numpy(image_parity.natural_frame().astype(np.float32)).to(dev)
    nat = torch.cat([nat[..., :3], torch.ones_like(nat[..., :1])], dim=-1).half().contiguous() if nat.shape[-1] >= 3 else nat
    for K in (32, 48, 56, 64):
        doc["bench frame 1080p->4K, 64x16 tiles, K=%d" % K] = census(synth, 3840, 2160, K, 64, 16, 1)
    doc["bench frame 1080p->4K, 64x32 tiles, K=48"] = census(synth, 3840, 2160, 48, 64, 32, 1)
    doc["bench frame 1080p->4K, 64x32 tiles, K=56"] = census(synth, 3840, 2160, 56, 64, 32, 1)
    doc["bench frame 1440p->4K, 64x32 tiles, K=48"] = census(synth1440, 3840, 2160, 48, 64, 32, 0)
    nh, nw = nat.shape[:2]
    doc["natural %dx%d 2x, 64x16 tiles, K=48" % (nw, nh)] = census(nat, 2 * nw, 2 * nh, 48, 64, 16, 1)
    doc["natural %dx%d 2x, 64x32 tiles, K=48" % (nw, nh)] = census(nat, 2 * nw, 2 * nh, 48, 64, 32, 1)
    for k, v in doc.items():
        print(k, json.dumps(v), flush=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(doc, open(os.path.join(ROOT, "gpurun_out", "r06_strict_flag_census.json"), "w"), indent=1)
