"""TEST INFRASTRUCTURE — image-level distance of a GPU EASU -> RCAS result from the reference CHAIN.

The per-stage parity tests judge RCAS on the GPU's own intermediary (RCAS has a gain of 4-7x on
1-ULP input differences).  This module measures what an integrator sees instead: the final image of the product's
pipeline against the reference's own two passes chained on the same input,

    FsrEasuF (ffx-fsr/ffx_fsr1.h:315-437)  ->  round to the intermediary's storage (RTNE binary16)  ->  FsrRcasF (:684-769)

evaluated by `oracle/_ref` (the reference headers compiled verbatim) when it travelled, else by the plain-C restatement.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import numpy as np

# the true-ratio presets of the reference's sample (sample/src/DX12/FSRSample.h:79-95, PDF p.10) next to BASELINE's shapes
SHAPES = {
    "540p_to_1080p": (960, 540, 1920, 1080),       # BASELINE configs[0] shape, 2.0x
    "1080p_to_4k": (1920, 1080, 3840, 2160),       # configs[1], [3]: "Performance" 2.0x
    "1440p_to_4k": (2560, 1440, 3840, 2160),       # configs[2]: "Quality" 1.5x
    "4k_to_8k": (3840, 2160, 7680, 4320),          # configs[4]
    "831p_to_1080p": (1477, 831, 1920, 1080),      # "Ultra Quality" 1.3x at a 1080p target
    "1270p_to_4k": (2259, 1270, 3840, 2160),       # "Balanced" 1.7x
    "1662p_to_4k": (2954, 1662, 3840, 2160),       # "Ultra Quality" 1.3x
}


def natural_frame(width=1477, height=831, x0=0, y0=0, dtype=np.float16):
    """A (height, width, 4) crop of the natural-content fixture tests/golden/natural_1477x831_rgb8.npz (the top-left corner of
    the reference's screenshot.png: GUI text, sky gradient, foliage, texture; made by tests/golden/gen_natural.py): 8-bit code
    / 255 rounded to binary16, alpha = 1 — every value is binary16-representable, so the CPU checker (float32) and the GPU
    (fp16 storage) read identical inputs."""
    import os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "natural_1477x831_rgb8.npz")
    rgb = np.load(path)["rgb8"]
    if x0 < 0 or y0 < 0 or x0 + width > rgb.shape[1] or y0 + height > rgb.shape[0]:
        raise ValueError("crop %dx%d at (%d, %d) leaves the %dx%d fixture" % (width, height, x0, y0, rgb.shape[1], rgb.shape[0]))
    img = np.ones((height, width, 4), np.float32)
    img[..., :3] = rgb[y0:y0 + height, x0:x0 + width, :].astype(np.float32) / np.float32(255.0)
    h16 = img.astype(np.float16)
    return h16 if dtype == np.float16 else h16.astype(dtype)


def checker():
    import cpu_oracle
    return cpu_oracle.ref() if cpu_oracle.have_ref() else cpu_oracle.port()


def reference_chain(o, img_f32, out_w, out_h, sharpness=0.25, return_mid=False):
    """The reference's two passes chained: FsrEasuF -> RTNE binary16 (the RGBA16F intermediary) -> FsrRcasF.  float32 images
    holding binary16-representable input; returns the float32 result (to be rounded to binary16 by the comparison)."""
    h, w, _ = img_f32.shape
    con = o.FsrEasuCon(w, h, w, h, out_w, out_h)
    rc = o.FsrRcasCon(sharpness)
    mid = o.easu_f(np.ascontiguousarray(img_f32, np.float32), out_w, out_h, con).astype(np.float16).astype(np.float32)
    out = o.rcas_f(mid, rc)
    return (out, mid) if return_mid else out


def _mono(t):
    """binary16 bit patterns -> integers monotone in the value, on the device"""
    import torch
    i = t.view(torch.int16).to(torch.int32)
    return torch.where(i < 0, -(i & 0x7FFF), i)


def ulp_histogram(got16, want_f32):
    """got16: CUDA float16 tensor (H, W, 4); want_f32: numpy float32 (H, W, 4), rounded RTNE to binary16 on the device.
    Only R, G, B are counted (alpha is the constant 1 on both sides and would flatter every fraction by a quarter); alpha is
    checked for equality separately.  Returns a dict of plain numbers: max ULP, the histogram 0 / 1 / 2 / 3-4 / > 4 binary16
    ULP, the fractions bit-equal and within 1 ULP, NaN counts."""
    import torch
    want16 = torch.from_numpy(np.ascontiguousarray(want_f32, np.float32)).to(got16.device).to(torch.float16)
    alpha_equal = bool(torch.equal(got16[..., 3].view(torch.int16), want16[..., 3].view(torch.int16)))
    got16, want16 = got16[..., :3], want16[..., :3]
    both_nan = torch.isnan(got16) & torch.isnan(want16)
    d = (_mono(got16) - _mono(want16)).abs()
    d = torch.where(both_nan, torch.zeros_like(d), d)
    n = d.numel()
    c0 = int((d == 0).sum())
    c1 = int((d == 1).sum())
    c2 = int((d == 2).sum())
    c4 = int(((d > 2) & (d <= 4)).sum())
    cg = int((d > 4).sum())
    worst = int(d.max())
    res = {
        "values": n, "max_ulp": worst,
        "hist": {"0": c0, "1": c1, "2": c2, "3-4": c4, ">4": cg},
        "frac_bit_equal": round(c0 / n, 6), "frac_within_1ulp": round((c0 + c1) / n, 6),
        "nan_in_output": int(torch.isnan(got16).sum()), "nan_in_reference": int(torch.isnan(want16).sum()),
        "alpha_equal": alpha_equal,
    }
    if worst > 1:
        # where the worst value is and what it is: the 2+ ULP values sit where RCAS's limiter amplifies a 1-ULP
        # intermediary difference (the lobe is a ratio of small differences)
        idx = int(torch.argmax(d))
        y, rem = divmod(idx, d.shape[1] * d.shape[2])
        x, c = divmod(rem, d.shape[2])
        res["worst_at"] = {"x": x, "y": y, "channel": c, "got": float(got16[y, x, c]), "want": float(want16[y, x, c])}
    return res
