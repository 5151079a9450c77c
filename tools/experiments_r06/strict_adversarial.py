"""Round 6: an ADVERSARIAL search against the bound F-strict's threshold rests on (strict_stress.py samples content at random; this tool
looks for the worst content on purpose).

The bound: d = |default - EXACT| / (2^-24 M) <= kEasuStrictK (48 when this tool was written, 56 since its result) for every value EASU produces, M = the largest |R|,|G|,|B| among the pixel's
12 taps (include/fsr1_device_easu.hpp, "F-strict").  Random content over 8.2e12 values shows max d = 30.0.  Here the input image is a POPULATION
of T x T-texel tiles (T a multiple of the ratio's period, so a tile means the same at every tile position); a tile's fitness is the largest d
among the output pixels whose 12-tap window lies inside it; every generation the best quarter survives and the rest are replaced by mutated
copies of survivors (binary16-ULP nudges of texel channels, random texels, copies of neighbours, rescaling, row / column crossover, flips) —
a (mu + lambda) evolution strategy run by the GPU on ~14 000 tiles at once, per ratio.  Every `--check-every` generations the same image is also
sent through easu(STRICT) and easu(EXACT) in RGBA16F storage: a differing stored value would be the counter-example itself.

Writes gpurun_out/r06_strict_adversarial.json: per ratio the best d per generation (sampled), the worst tile found (binary16 bit patterns,
position phase), values evaluated, and the strict-vs-EXACT count.
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
fsr = importlib.import_module("fidelityfx-fsr_amd")

dev = "cuda"

# (ratio as out/in numerator, denominator, tile edge in texels = a multiple of the denominator's period)
CONFIGS = [(2, 1, 8), (3, 2, 8), (13, 10, 10), (17, 10, 10), (3, 1, 8), (5, 4, 8), (19, 10, 10), (4, 3, 9),
           (4, 1, 8), (5, 2, 8), (1, 1, 8), (3, 4, 8)]  # (beyond the presets: 4x, 2.5x, 1x, a minification)


def tiles_of(img, T):
    ih, iw, c = img.shape
    return img.view(ih // T, T, iw // T, T, c).permute(0, 2, 1, 3, 4).reshape(-1, T, T, c)


def image_of(tiles, ny, nx):
    n, T, _, c = tiles.shape
    return tiles.view(ny, nx, T, T, c).permute(0, 2, 1, 3, 4).reshape(ny * T, nx * T, c).contiguous()


def random_tiles(n, T, g):
    """Fresh tiles of several kinds (binary16 bit patterns as int16, non-negative finite values)."""
    r = lambda *s: torch.rand(*s, device=dev, generator=g)
    kind = torch.randint(0, 7, (n, 1, 1, 1), device=dev, generator=g)
    u = r(n, T, T, 3)
    v = torch.where(kind == 0, u, torch.zeros_like(u))
    v = torch.where(kind == 1, u ** 6, v)                                                  # dark
    v = torch.where(kind == 2, (u > 0.5).float() * r(n, 1, 1, 3) + 0.02 * r(n, 1, 1, 3), v)  # two-level
    v = torch.where(kind == 3, torch.exp((u - 0.5) * 12.0), v)                              # HDR-ish magnitudes
    ramp = torch.linspace(0, 1, T, device=dev)
    grad = (ramp[None, :, None, None] * r(n, 1, 1, 3) + ramp[None, None, :, None] * r(n, 1, 1, 3)) * 0.5 + 0.05 * u
    v = torch.where(kind == 4, grad, v)
    v = torch.where(kind == 5, (r(n, T, T, 1) > 0.8).float() * r(n, 1, 1, 3), v)            # sparse bright texels
    # on / off patterns of HDR range (what the second search's worst tiles look like): each channel of each texel either `lo` or `hi`
    v = torch.where(kind == 6, torch.where(u > r(n, 1, 1, 1), 100.0 + 400.0 * r(n, 1, 1, 3), 0.01 + 0.05 * r(n, 1, 1, 3)), v)
    return v.clamp(0, 60000.0).half()


def mutate(parents, g, T):
    """One mutation per child, chosen at random; operates on binary16 values (bit patterns for the ULP nudges)."""
    n = parents.shape[0]
    c = parents.clone()
    op = torch.randint(0, 9, (n,), device=dev, generator=g)
    bits = c.view(torch.int16)
    # 0, 1: nudge k random texel channels by a few binary16 ULPs
    k_mask = torch.rand(n, T, T, 3, device=dev, generator=g) < (torch.rand(n, 1, 1, 1, device=dev, generator=g) * 0.15 + 0.01)
    delta = torch.randint(-4, 5, (n, T, T, 3), device=dev, generator=g, dtype=torch.int32)
    nudged = (bits.to(torch.int32) + delta.to(torch.int32)).clamp(0, 0x7BFF).to(torch.int16)
    sel = ((op <= 1)[:, None, None, None]) & k_mask
    bits.copy_(torch.where(sel, nudged, bits))
    # 2: replace a few texels with random values of the tile's own magnitude
    mag = c.float().amax(dim=(1, 2, 3), keepdim=True).clamp_min(1e-4)
    t_mask = torch.rand(n, T, T, 1, device=dev, generator=g) < 0.05
    rnd = (torch.rand(n, T, T, 3, device=dev, generator=g) * mag).half()
    c.copy_(torch.where(((op == 2)[:, None, None, None]) & t_mask, rnd, c))
    # 3: a texel takes its left / upper neighbour's value (flat runs, edges)
    shifted = torch.where(torch.rand(n, 1, 1, 1, device=dev, generator=g) < 0.5, torch.roll(c, 1, dims=1), torch.roll(c, 1, dims=2))
    n_mask = torch.rand(n, T, T, 1, device=dev, generator=g) < 0.1
    c.copy_(torch.where(((op == 3)[:, None, None, None]) & n_mask, shifted, c))
    # 4: rescale the tile (not by a power of two: the rounding pattern moves)
    f = (0.5 + 1.5 * torch.rand(n, 1, 1, 1, device=dev, generator=g))
    c.copy_(torch.where((op == 4)[:, None, None, None], (c.float() * f).clamp(0, 60000.0).half(), c))
    # 5: crossover with another child's parent: rows below a random cut
    other = c[torch.randperm(n, device=dev, generator=g)]
    cut = torch.randint(1, T, (n, 1, 1, 1), device=dev, generator=g)
    rows = torch.arange(T, device=dev)[None, :, None, None]
    c.copy_(torch.where(((op == 5)[:, None, None, None]) & (rows >= cut), other, c))
    # 6: one channel of a few texels (colourful edges)
    ch = torch.randint(0, 3, (n, 1, 1, 1), device=dev, generator=g) == torch.arange(3, device=dev)[None, None, None, :]
    c.copy_(torch.where(((op == 6)[:, None, None, None]) & t_mask & ch, rnd, c))
    # 7: nudge EVERY channel of one texel by +-1 ULP (fine search around a good tile)
    one = (torch.rand(n, T, T, 1, device=dev, generator=g) < 1.5 / (T * T))
    bits = c.view(torch.int16)
    pm = (torch.randint(0, 2, (n, T, T, 3), device=dev, generator=g, dtype=torch.int32) * 2 - 1)
    bits.copy_(torch.where(((op == 7)[:, None, None, None]) & one, (bits.to(torch.int32) + pm).clamp(0, 0x7BFF).to(torch.int16), bits))
    # 8: flip a few texel channels between the tile's smallest and largest value (on / off patterns: the sign pattern of the weight errors)
    cf = c.float()
    lo, hi = cf.amin(dim=(1, 2, 3), keepdim=True), cf.amax(dim=(1, 2, 3), keepdim=True)
    f_mask = torch.rand(n, T, T, 3, device=dev, generator=g) < 0.03
    flipped = torch.where(cf > 0.5 * (lo + hi), lo, hi).half()
    c.copy_(torch.where(((op == 8)[:, None, None, None]) & f_mask, flipped, c))
    return c


def search(num, den, T, seconds, seed, check_every, islands=False):
    g = torch.Generator(device=dev).manual_seed(seed * 1000 + num * 10 + den)
    nx, ny = 1280 // T, 720 // T
    iw, ih = nx * T, ny * T
    assert (iw * num) % den == 0 and (ih * num) % den == 0
    ow, oh = iw * num // den, ih * num // den
    con = fsr.FsrEasuCon(iw, ih, iw, ih, ow, oh)
    c = np.asarray(con, np.uint32).view(np.float32)
    fx = torch.floor(torch.arange(ow, device=dev, dtype=torch.float32) * float(c[0]) + float(c[2])).to(torch.int64)
    fy = torch.floor(torch.arange(oh, device=dev, dtype=torch.float32) * float(c[1]) + float(c[3])).to(torch.int64)
    own_x = ((fx - 1) >= 0) & ((fx + 2) < iw) & (torch.div(fx - 1, T, rounding_mode="floor") == torch.div(fx + 2, T, rounding_mode="floor"))
    own_y = ((fy - 1) >= 0) & ((fy + 2) < ih) & (torch.div(fy - 1, T, rounding_mode="floor") == torch.div(fy + 2, T, rounding_mode="floor"))
    nt = nx * ny
    tile_id = torch.where(own_y[:, None] & own_x[None, :], torch.div(fy, T, rounding_mode="floor")[:, None] * nx + torch.div(fx, T, rounding_mode="floor")[None, :],
                          torch.full((1, 1), nt, device=dev, dtype=torch.int64)).flatten()
    owned_values = int((tile_id < nt).sum()) * 3
    owned_per_tile = torch.zeros(nt + 1, device=dev).scatter_add(0, tile_id, torch.ones(tile_id.shape[0], device=dev))[:nt].clamp_min(1.0)
    gx = [(fx + dx).clamp(0, iw - 1) for dx in (-1, 0, 1, 2)]
    gy = [(fy + dy).clamp(0, ih - 1) for dy in (-1, 0, 1, 2)]

    pop = random_tiles(nt, T, g)
    out_d, out_e = (torch.empty(oh, ow, 4, dtype=torch.float32, device=dev) for _ in range(2))
    ex16, st16 = (torch.empty(oh, ow, 4, dtype=torch.float16, device=dev) for _ in range(2))
    best, best_tile, best_where = 0.0, None, None
    trace, values, differing, checked = [], 0, 0, 0
    gen, t0 = 0, time.perf_counter()
    elite = nt // 4
    while time.perf_counter() - t0 < seconds:
        img16 = torch.cat([image_of(pop, ny, nx), torch.ones(ih, iw, 1, dtype=torch.float16, device=dev)], dim=-1).contiguous()
        s32 = img16.float().contiguous()
        fsr.easu(s32, out_d, con=con)
        fsr.easu(s32, out_e, con=con, flags=fsr.FLAG_MATH_EXACT)
        dv, ev = out_d[..., :3], out_e[..., :3]
        fin = torch.isfinite(dv) & torch.isfinite(ev)
        delta = torch.where(fin, (dv.double() - ev.double()).abs(), torch.zeros((), device=dev, dtype=torch.float64)).amax(dim=-1)
        mag = s32[..., :3].abs().amax(dim=-1)
        M = torch.zeros(oh, ow, device=dev)
        for iy, dxs in ((0, (1, 2)), (1, (0, 1, 2, 3)), (2, (0, 1, 2, 3)), (3, (1, 2))):
            rowsel = mag[gy[iy]]
            for ix in dxs:
                M = torch.maximum(M, rowsel[:, gx[ix]])
        rr = (delta / (M.double().clamp_min(2.0 ** -126) * 2.0 ** -24)).float().flatten()
        fit = torch.zeros(nt + 1, device=dev).scatter_reduce(0, tile_id, rr, "amax", include_self=True)[:nt]
        values += owned_values
        top = float(fit.max())
        if top > best:
            i = int(fit.argmax())
            best, best_tile = top, pop[i].clone()
            best_where = {"generation": gen, "tile": [i % nx, i // nx]}
        if gen % 25 == 0:
            trace.append([gen, round(top, 2), round(float(fit.median()), 2)])
        if islands:
            # selection pressure: the islands of the upper half of the image climb the tile's MEAN d (the regime: where the two arithmetics are
            # systematically far apart and the tail of the rounding noise is heaviest), those of the lower half its maximum
            mean = torch.zeros(nt + 1, device=dev).scatter_add(0, tile_id, rr)[:nt] / owned_per_tile
            sel_fit = torch.where(torch.arange(nt, device=dev) < nt // 2, 4.0 * mean + 0.25 * fit, fit)
        else:
            sel_fit = fit
        if check_every and gen % check_every == 0:
            fsr.easu(img16, ex16, con=con, flags=fsr.FLAG_MATH_EXACT)
            fsr.easu(img16, st16, con=con, flags=fsr.FLAG_MATH_STRICT)
            bad = (ex16.view(torch.int16) != st16.view(torch.int16)) & ~(torch.isnan(ex16) & torch.isnan(st16))
            differing += int(bad.sum())
            checked += ex16.numel()
        # (mu + lambda): the best quarter stays, the rest become mutated copies of survivors (better ones more often); a few fresh tiles
        if islands:
            # one island per row of tiles: selection inside the row (the rows keep different lineages: no collapse onto one tile); every 40
            # generations each row's best tile migrates to the next row
            order = torch.argsort(sel_fit.view(ny, nx), dim=1, descending=True)          # [ny, nx] tile columns, best first
            e_row = nx // 4
            pick = (torch.rand(ny, nx - e_row, device=dev, generator=g) ** 2 * e_row).long()
            parents_col = torch.gather(order, 1, pick)
            row_base = (torch.arange(ny, device=dev) * nx)[:, None]
            children = mutate(pop[(row_base + parents_col).flatten()], g, T)
            fresh = torch.rand(children.shape[0], device=dev, generator=g) < 0.02
            children = torch.where(fresh[:, None, None, None], random_tiles(children.shape[0], T, g), children)
            # in place: a survivor keeps its tile POSITION (off 2x the sub-texel position is rounded from the absolute coordinate, so a tile's d
            # belongs to the tile at that position: moved, it would be drawn again); children take the places of the row's losers
            losers = (row_base + order[:, e_row:]).flatten()
            pop[losers] = children
            if gen % 40 == 39:
                pop[(row_base[:, 0] + order[:, -1])] = torch.roll(pop[(row_base[:, 0] + order[:, 0])], 1, dims=0)
        else:
            order = torch.argsort(sel_fit, descending=True)
            parents_idx = order[(torch.rand(nt - elite, device=dev, generator=g) ** 2 * elite).long()]
            children = mutate(pop[parents_idx], g, T)
            fresh = torch.rand(nt - elite, device=dev, generator=g) < 0.02
            children = torch.where(fresh[:, None, None, None], random_tiles(nt - elite, T, g), children)
            pop[order[elite:]] = children
        gen += 1
    return {"ratio": "%d/%d" % (num, den), "in": [iw, ih], "out": [ow, oh], "tile_texels": T, "tiles": nt, "generations": gen, "islands": islands,
            "values_evaluated": values, "max_d": round(best, 3), "found": best_where,
            "worst_tile_rgb_binary16_bits": None if best_tile is None else best_tile.view(torch.int16).to(torch.int32).cpu().numpy().astype(np.uint16).tolist(),
            "trace_generation_best_median": trace, "strict_vs_exact_values_checked": checked, "strict_vs_exact_differing": differing,
            "seconds": round(time.perf_counter() - t0, 1)}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120.0, help="per ratio")
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--check-every", type=int, default=5)
    ap.add_argument("--ratios", default="", help="comma-separated subset of CONFIGS indices")
    ap.add_argument("--islands", action="store_true", help="one island per row of tiles (no collapse onto one lineage); half of them select on the tile's mean d")
    ap.add_argument("--out", default="r06_strict_adversarial.json")
    args = ap.parse_args()
    fsr.load()
    which = [int(x) for x in args.ratios.split(",")] if args.ratios else range(len(CONFIGS))
    runs = []
    for k in which:
        num, den, T = CONFIGS[k]
        r = search(num, den, T, args.seconds, args.seed, args.check_every, args.islands)
        print(json.dumps({kk: v for kk, v in r.items() if kk not in ("worst_tile_rgb_binary16_bits", "trace_generation_best_median")}), flush=True)
        runs.append(r)
    doc = {"what": "evolutionary search for the largest d = |default - EXACT| / (2^-24 M) of EASU (F-strict's threshold: 56, 48 before this search; random content: max 30.0 of 8.2e12 values)",
           "threshold": 56, "seed": args.seed, "islands": args.islands, "max_d": max(r["max_d"] for r in runs), "strict_vs_exact_differing": sum(r["strict_vs_exact_differing"] for r in runs),
           "strict_vs_exact_values_checked": sum(r["strict_vs_exact_values_checked"] for r in runs), "runs": runs}
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", args.out), "w") as f:
        json.dump(doc, f)
    print(json.dumps({"max_d": doc["max_d"], "differing": doc["strict_vs_exact_differing"], "checked": doc["strict_vs_exact_values_checked"]}))
