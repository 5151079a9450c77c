"""Round 6 (CPU, numpy): where F-strict's distance d comes from, and how far twelve weight errors could add up.

Restates, in binary32 arithmetic, the two ways the product computes an EASU tap weight from the same pixel terms —
  * the reference's operation order (include/fsr1_device_easu.hpp easu_filter<true>; ffx_fsr1.h:250-281, :397-409), and
  * the default arithmetic's Q-form / Horner evaluation in u = d2 / clp (easu_tap_terms, easu_tap_weight) —
for random directions, edge lengths and sub-texel positions, and reports, in units of 2^-24:
  (1) the per-tap weight difference dw = w_default - w_reference, by clp (the window's clip point: 2 .. 4.76);
  (2) per pixel, what an adversary who could set every tap's colour to 0 or M independently of the direction analysis would reach:
      |dx| / M = | sum_on dw_i - (sum_on w_i / W) sum_all dw_i | / W  with  on = {i : dw_i > 0}  (first order in dw),
      against the same sum with random colours;
  (3) per tap, against the TRUE weight (binary64): the reference order's own rounding error, the default arithmetic's, and the default's two
      parts (its u, its polynomial) — the distance F-strict has to cover is the sum of two independent rounding noises of the same size.
fma(a, b, c) is emulated as float32(float64(a) * float64(b) + float64(c)) (the product is exact in binary64; the double rounding is
rare and does not matter to a maximum over 1e7 samples); v_rcp_f32 (1 ULP) as float32(1 / x).  A model of the arithmetic, not of the
kernels: the measured d of the real kernels is tools/experiments_r06/strict_stress.py / strict_adversarial.py.
"""
import json
import sys

import numpy as np

f32 = np.float32


def fma(a, b, c):
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(f32)


def aprx_lo_rcp(a):  # ffx_a.h APrxLoRcpF1: AF1_AU1(AU1_(0x7ef07ebb) - AU1_AF1(a))
    return (np.uint32(0x7EF07EBB) - a.view(np.uint32)).view(f32)


def run(n, seed):
    g = np.random.default_rng(seed)
    theta = g.uniform(0, 2 * np.pi, n)
    norm = g.uniform(0.94, 1.06, n)  # APrxLoRsqF1 leaves the direction a few per cent off unit length
    dirx, diry = (np.cos(theta) * norm).astype(f32), (np.sin(theta) * norm).astype(f32)
    ln = g.uniform(0, 1, n).astype(f32) ** f32(0.5)  # len after `len = len * 0.5; len *= len`, in [0, 1], edges (len -> 1) favoured
    ppx, ppy = g.uniform(0, 1, n).astype(f32), g.uniform(0, 1, n).astype(f32)
    one, c_lob = f32(1.0), f32((1.0 / 4.0 - 0.04) - 0.5)
    # shape, reference order (:397-409)
    st_e = (dirx * dirx + diry * diry) * aprx_lo_rcp(np.maximum(np.abs(dirx), np.abs(diry)))
    l2x_e = (st_e - one) * ln + one
    l2y_e = f32(-0.5) * ln + one
    lob_e = c_lob * ln + f32(0.5)
    clp_e = aprx_lo_rcp(lob_e)
    # shape, default arithmetic (mad<false> = fma)
    st_d = fma(dirx, dirx, diry * diry) * aprx_lo_rcp(np.maximum(np.abs(dirx), np.abs(diry)))
    l2x_d = fma(st_d - one, ln, np.full(n, one))
    l2y_d = fma(np.full(n, f32(-0.5)), ln, np.full(n, one))
    lob_d = fma(np.full(n, c_lob), ln, np.full(n, f32(0.5)))
    clp_d = aprx_lo_rcp(lob_d)
    rclp = (one / clp_d).astype(f32)
    sx, sy = l2x_d * l2x_d * rclp, l2y_d * l2y_d * rclp
    dxx, dyy, dxy2 = dirx * dirx, diry * diry, f32(2.0) * (dirx * diry)
    q00, q11, q01 = fma(dxx, sx, dyy * sy), fma(dyy, sx, dxx * sy), dxy2 * (sx - sy)
    k2, k1, k3 = f32(0.25) * clp_d * clp_d, f32(-1.25) * clp_d, lob_d * clp_d
    taps = [(0, -1), (1, -1), (-1, 1), (0, 1), (0, 0), (-1, 0), (1, 1), (2, 1), (2, 0), (1, 0), (1, 2), (0, 2)]  # b c i j f e k l h g o n
    dws, wes = [], []
    for tx, ty in taps:
        ox, oy = f32(tx) - ppx, f32(ty) - ppy
        # reference order
        vx = (ox * dirx) + (oy * diry)
        vy = (ox * (-diry)) + (oy * dirx)
        vx = vx * l2x_e
        vy = vy * l2y_e
        d2 = np.minimum(vx * vx + vy * vy, clp_e)
        wb = f32(2.0 / 5.0) * d2 + f32(-1.0)
        wa = lob_e * d2 + f32(-1.0)
        wb = wb * wb
        wa = wa * wa
        wb = f32(25.0 / 16.0) * wb + f32(-(25.0 / 16.0 - 1.0))
        we = wb * wa
        # default arithmetic
        s, b = q01 * oy, q11 * (oy * oy)
        u = np.clip(fma(ox, fma(q00, ox, s), b), f32(0), f32(1))
        base = fma(fma(k2, u, k1), u, np.full(n, one))
        wad = fma(k3, u, np.full(n, f32(-1.0)))
        wd = base * (wad * wad)
        dws.append((wd.astype(np.float64) - we.astype(np.float64)) * 2.0 ** 24)
        wes.append(we.astype(np.float64))
    dw, we = np.stack(dws), np.stack(wes)  # [12, n]
    W = we.sum(axis=0)
    ok = W > 0.5  # (the kernels see 0.65 .. 2.65: drop the model's few outliers of the sampling)
    out = {"samples": int(n), "kept": int(ok.sum()), "W_min_max": [round(float(W[ok].min()), 3), round(float(W[ok].max()), 3)]}
    a = np.abs(dw[:, ok])
    clp = clp_d[ok]
    out["per_tap_abs_dw_units"] = {"mean": round(float(a.mean()), 3), "p99": round(float(np.percentile(a, 99)), 2), "max": round(float(a.max()), 2)}
    bins = [(1.9, 2.5), (2.5, 3.2), (3.2, 4.0), (4.0, 5.0)]
    out["per_tap_abs_dw_by_clp"] = {"%.1f-%.1f" % b: {"mean": round(float(a[:, (clp >= b[0]) & (clp < b[1])].mean()), 3), "max": round(float(a[:, (clp >= b[0]) & (clp < b[1])].max()), 2)} for b in bins}
    dwk, wk, Wk = dw[:, ok], we[:, ok], W[ok]
    on = dwk > 0
    adv = np.abs((dwk * on).sum(axis=0) - ((wk * on).sum(axis=0) / Wk) * dwk.sum(axis=0)) / Wk
    off = ~on
    adv2 = np.abs((dwk * off).sum(axis=0) - ((wk * off).sum(axis=0) / Wk) * dwk.sum(axis=0)) / Wk
    adv = np.maximum(adv, adv2)
    col = g.uniform(0, 1, dwk.shape)
    x = (col * wk).sum(axis=0) / Wk
    rnd = np.abs(((col - x) * dwk).sum(axis=0)) / Wk
    full = a.sum(axis=0) / Wk
    for name, v in (("on_off_adversary_units", adv), ("random_colours_units", rnd), ("sum_of_abs_dw_over_W_units", full)):
        out[name] = {"mean": round(float(v.mean()), 2), "p99.9": round(float(np.percentile(v, 99.9)), 1), "max": round(float(v.max()), 1)}
    return out


def decompose(n, seed):
    """Per tap, against the TRUE weight (the reference's formula evaluated in binary64 on the same binary32 pixel terms): how far the reference
    order's binary32 weight is from it, how far the default arithmetic's is, and the default's two parts — its u = d2 / clp (Q-form) and its
    polynomial in u (Horner) — each with the other part evaluated in binary64."""
    g = np.random.default_rng(seed)
    f64 = np.float64
    theta = g.uniform(0, 2 * np.pi, n)
    norm = g.uniform(0.94, 1.06, n)
    dirx, diry = (np.cos(theta) * norm).astype(f32), (np.sin(theta) * norm).astype(f32)
    ln = g.uniform(0, 1, n).astype(f32) ** f32(0.5)
    ppx, ppy = g.uniform(0, 1, n).astype(f32), g.uniform(0, 1, n).astype(f32)
    one, c_lob = f32(1.0), f32((1.0 / 4.0 - 0.04) - 0.5)
    full = lambda v: np.full(n, f32(v))  # noqa: E731
    st_e = (dirx * dirx + diry * diry) * aprx_lo_rcp(np.maximum(np.abs(dirx), np.abs(diry)))
    l2x_e, l2y_e, lob_e = (st_e - one) * ln + one, f32(-0.5) * ln + one, c_lob * ln + f32(0.5)
    clp_e = aprx_lo_rcp(lob_e)
    st_d = fma(dirx, dirx, diry * diry) * aprx_lo_rcp(np.maximum(np.abs(dirx), np.abs(diry)))
    l2x_d, l2y_d, lob_d = fma(st_d - one, ln, full(1)), fma(full(-0.5), ln, full(1)), fma(full(c_lob), ln, full(0.5))
    clp_d = aprx_lo_rcp(lob_d)
    rclp = (one / clp_d).astype(f32)
    sx, sy = l2x_d * l2x_d * rclp, l2y_d * l2y_d * rclp
    dxx, dyy, dxy2 = dirx * dirx, diry * diry, f32(2.0) * (dirx * diry)
    q00, q11, q01 = fma(dxx, sx, dyy * sy), fma(dyy, sx, dxx * sy), dxy2 * (sx - sy)
    k2, k1, k3 = f32(0.25) * clp_d * clp_d, f32(-1.25) * clp_d, lob_d * clp_d
    taps = [(0, -1), (1, -1), (-1, 1), (0, 1), (0, 0), (-1, 0), (1, 1), (2, 1), (2, 0), (1, 0), (1, 2), (0, 2)]
    res = {k: [] for k in ("reference_vs_true", "default_vs_true", "default_vs_reference", "default_u_part", "default_polynomial_part")}
    D = lambda a: a.astype(f64)  # noqa: E731
    for tx, ty in taps:
        ox, oy = f32(tx) - ppx, f32(ty) - ppy
        vx = (D(ox) * D(dirx) + D(oy) * D(diry)) * D(l2x_e)
        vy = (D(ox) * (-D(diry)) + D(oy) * D(dirx)) * D(l2y_e)
        d2t = np.minimum(vx * vx + vy * vy, D(clp_e))
        wt = (25.0 / 16.0 * (D(f32(0.4)) * d2t - 1) ** 2 - (25.0 / 16.0 - 1)) * (D(lob_e) * d2t - 1) ** 2
        vxf, vyf = ((ox * dirx) + (oy * diry)) * l2x_e, ((ox * (-diry)) + (oy * dirx)) * l2y_e
        d2 = np.minimum(vxf * vxf + vyf * vyf, clp_e)
        wb, wa = f32(0.4) * d2 + f32(-1), lob_e * d2 + f32(-1)
        we = (f32(25.0 / 16.0) * (wb * wb) + f32(-(25.0 / 16.0 - 1.0))) * (wa * wa)
        s, b = q01 * oy, q11 * (oy * oy)
        u = np.clip(fma(ox, fma(q00, ox, s), b), f32(0), f32(1))
        wad = fma(k3, u, full(-1))
        wd = fma(fma(k2, u, k1), u, full(1)) * (wad * wad)
        ut = np.clip(d2t / D(clp_d), 0, 1)
        P = lambda uu: (D(k2) * uu * uu + D(k1) * uu + 1) * (D(k3) * uu - 1) ** 2  # noqa: E731
        utr = ut.astype(f32)
        wad2 = fma(k3, utr, full(-1))
        wp = fma(fma(k2, utr, k1), utr, full(1)) * (wad2 * wad2)
        S = 2.0 ** 24
        res["reference_vs_true"].append(np.abs(D(we) - wt) * S)
        res["default_vs_true"].append(np.abs(D(wd) - wt) * S)
        res["default_vs_reference"].append(np.abs(D(wd) - D(we)) * S)
        res["default_u_part"].append(np.abs(P(D(u)) - P(ut)) * S)
        res["default_polynomial_part"].append(np.abs(D(wp) - P(D(utr))) * S)
    out = {k: {"mean": round(float(np.mean(v)), 3), "p99.9": round(float(np.percentile(np.stack(v), 99.9)), 2), "max": round(float(np.max(v)), 2)} for k, v in res.items()}
    out["shape_terms_bit_identical_between_the_two"] = round(float(np.mean((l2x_d == l2x_e) & (l2y_d == l2y_e) & (lob_d == lob_e))), 3)
    return out


if __name__ == "__main__":
    n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 2_000_000
    res = [run(n, s) for s in (1, 2, 3, 4, 5)]
    agg = {"model": "binary32 restatement of the two tap-weight evaluations; units of 2^-24 (x the window's magnitude M for the sums)", "runs": res,
           "per_tap_decomposition_against_the_true_weight": decompose(min(n, 1_000_000), 9)}
    print(json.dumps(agg, indent=1))
