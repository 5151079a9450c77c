"""Deterministic synthetic frames for parity tests and the bench (no files, no network).

Content follows SURVEY.md §8(d): diagonal hard-edge stripes + per-channel smooth sinusoids +
low-amplitude hash noise, plus blocks that exercise the RCAS/EASU corner cases (all-black,
all-white, saturated primaries, a 1-pixel checkerboard).  Values lie in [0,1], alpha = 1, and the
frame is quantised to binary16 *before* anyone sees it, so the CPU oracle (float32 holding fp16
values) and the GPU kernels (fp16 storage) read identical inputs.
"""
import numpy as np


def _hash_noise(x, y, seed):
    """uint32 integer hash -> [0,1) float32, vectorised; one LCG step per mix (s*1664525+1013904223)."""
    s = (x.astype(np.uint64) * 73856093) ^ (y.astype(np.uint64) * 19349663) ^ np.uint64(seed & 0xFFFFFFFF)
    s &= 0xFFFFFFFF
    for _ in range(2):
        s = (s * 1664525 + 1013904223) & 0xFFFFFFFF
        s ^= s >> 15
    return ((s >> 8).astype(np.float32)) * np.float32(1.0 / 16777216.0)


def synthetic_frame(width, height, k=0, blocks=True, dtype=np.float16):
    """Frame `k` of a batch: (height, width, 4) RGBA array in `dtype` (float16 or float32).

    The float32 variant holds the *same* fp16-representable values (it is the fp16 frame widened).
    """
    seed = (0x9E3779B9 * (k + 1)) & 0xFFFFFFFF
    y, x = np.meshgrid(np.arange(height, dtype=np.int64), np.arange(width, dtype=np.int64), indexing="ij")
    stripes = np.where(((x + 2 * y + k) % 16) < 8, np.float32(0.8), np.float32(0.1))
    fx = x.astype(np.float32)
    fy = y.astype(np.float32)
    img = np.empty((height, width, 4), np.float32)
    for c, (ax, ay, ph) in enumerate(((0.031, 0.017, 0.0), (0.013, 0.029, 1.3), (0.023, 0.011, 2.1))):
        smooth = 0.5 + 0.5 * np.sin(fx * np.float32(ax) + fy * np.float32(ay) + np.float32(ph + 0.37 * k))
        noise = _hash_noise(x, y, seed + 0x1234567 * (c + 1)) * np.float32(0.1)
        img[..., c] = 0.55 * stripes + 0.35 * smooth.astype(np.float32) + noise
    img[..., 3] = 1.0
    np.clip(img, 0.0, 1.0, out=img)
    if blocks:
        b = max(4, min(32, width // 8, height // 8))

        def put(ix, iy, rgb):
            x0, y0 = ix * b, iy * b
            if x0 + b <= width and y0 + b <= height:
                img[y0:y0 + b, x0:x0 + b, :3] = rgb

        put(1, 1, (0.0, 0.0, 0.0))
        put(2, 1, (1.0, 1.0, 1.0))
        put(3, 1, (1.0, 0.0, 0.0))
        put(4, 1, (0.0, 1.0, 0.0))
        put(5, 1, (0.0, 0.0, 1.0))
        x0, y0 = 6 * b, 1 * b
        if x0 + b <= width and y0 + b <= height:
            chk = (((x[y0:y0 + b, x0:x0 + b] + y[y0:y0 + b, x0:x0 + b]) & 1) == 0).astype(np.float32)
            img[y0:y0 + b, x0:x0 + b, :3] = chk[..., None]
    h16 = img.astype(np.float16)
    return h16 if dtype == np.float16 else h16.astype(dtype)


def kat_frame_64x36():
    """The 64x36 known-answer frame of SURVEY.md Appendix B.2 (sequential LCG, float32 values)."""
    W, H = 64, 36
    n = W * H * 2
    s = np.empty(n, np.uint64)
    cur = 12345
    for i in range(n):
        cur = (cur * 1664525 + 1013904223) & 0xFFFFFFFF
        s[i] = cur
    rnd = ((s >> 8).astype(np.float32) * np.float32(1.0 / 16777216.0)).reshape(H, W, 2)
    y, x = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    e = np.where(((x + 2 * y) % 16) < 8, np.float32(0.8), np.float32(0.1))
    img = np.empty((H, W, 4), np.float32)
    img[..., 0] = np.float32(0.5) * e + np.float32(0.5) * rnd[..., 0] * np.float32(0.2)
    img[..., 1] = e
    img[..., 2] = np.float32(0.25) + np.float32(0.5) * e * rnd[..., 1]
    img[..., 3] = 1.0
    return img


def adversarial_frame(width, height, k=0, dtype=np.float16):
    """A frame of finite but hostile binary16 values for the special-value parity tests: HDR highlights up to the top of
    the binary16 range, exact zeros, negative (scRGB) texels, binary16 subnormals and isolated single-texel spikes on top of
    `synthetic_frame`.  No Inf / NaN inputs: what the shading languages' min / max do with those is not defined by the
    reference; Inf and NaN that the arithmetic itself produces from these inputs are part of the comparison."""
    img = synthetic_frame(width, height, k=k, dtype=np.float32)
    y, x = np.meshgrid(np.arange(height, dtype=np.int64), np.arange(width, dtype=np.int64), indexing="ij")
    seed = (0x85EBCA6B * (k + 3)) & 0xFFFFFFFF
    u = _hash_noise(x, y, seed)                  # [0, 1)
    sel = _hash_noise(x // 3, y // 2, seed ^ 0x5bd1e995)  # small patches share a class
    gain = np.select([sel < 0.55, sel < 0.70, sel < 0.80], [1.0, 8.0, 512.0], 60000.0).astype(np.float32)
    img[..., :3] = np.minimum(img[..., :3] * gain[..., None], np.float32(65504.0))
    img[(u < 0.04), :3] = 0.0
    img[(u >= 0.04) & (u < 0.07), :3] *= -1.0
    img[(u >= 0.07) & (u < 0.10), :3] = np.float32(3.0e-6)      # binary16 subnormal
    img[(u >= 0.10) & (u < 0.12), :3] = np.float32(5.9604645e-8)  # the smallest one
    img[(u >= 0.12) & (u < 0.13), 0] = np.float32(65504.0)       # single-channel spikes
    img[(u >= 0.13) & (u < 0.14), 2] = np.float32(6.1035156e-5)  # smallest normal
    return img.astype(np.float16).astype(dtype)
