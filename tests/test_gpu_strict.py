"""F-strict (FSR1_FLAG_MATH_STRICT, round 6): the default arithmetic's speed with FsrEasuF's bits.

Contract under test (include/fsr1_hip.h):
  * EASU's stored image is BIT-IDENTICAL to FSR1_FLAG_MATH_EXACT's — which tests/test_gpu_fullframe.py pins to the CPU-evaluated
    FsrEasuF (ffx-fsr/ffx_fsr1.h:315-437) — for every storage format with a conversion (RGBA16F, RGBA8, R10G10B10A2), at every ratio,
    for ragged sizes, batches, bands, hostile values and for content that sends every pixel through the re-evaluation;
  * RCAS runs the default arithmetic under the flag, so every pipeline (two dispatches, fused launch in each of its launch shapes,
    fsr1_upscale, fsr1_pipeline) produces the bits of  easu(EXACT) -> rcas(default);
  * where EASU has no strict variant (RGBA32F storage, `c *= c`) the flag takes the EXACT kernels.
The image-level distance from the reference CHAIN is measured in tests/test_gpu_image_parity.py.
"""
import importlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
frames = importlib.import_module("fidelityfx-fsr_amd.frames")

SHAPES = {
    "540p_to_1080p": (960, 540, 1920, 1080),
    "1080p_to_4k": (1920, 1080, 3840, 2160),
    "1440p_to_4k": (2560, 1440, 3840, 2160),
    "4k_to_8k": (3840, 2160, 7680, 4320),
    "831p_to_1080p": (1477, 831, 1920, 1080),
    "1270p_to_4k": (2259, 1270, 3840, 2160),
    "1662p_to_4k": (2954, 1662, 3840, 2160),
    "ragged_2x": (333, 211, 666, 422),
    "ragged_1p7x": (333, 211, 567, 359),
    "tiny_2x": (5, 3, 10, 6),
    "one_texel": (1, 1, 3, 2),
    "minify": (640, 360, 480, 270),
}


def as_storage(img16, fmt):
    """fp16 RGBA frame -> CUDA tensor in the storage format `fmt` ("f16", "u8", "r10")."""
    if fmt == "f16":
        return torch.from_numpy(np.ascontiguousarray(img16)).cuda()
    x = np.clip(img16.astype(np.float32), 0.0, 1.0)
    if fmt == "u8":
        return torch.from_numpy(np.rint(x * 255.0).astype(np.uint8)).cuda()
    c = np.rint(x[..., :3] * 1023.0).astype(np.uint32)
    word = c[..., 0] | (c[..., 1] << 10) | (c[..., 2] << 20) | (np.uint32(3) << 30)
    return torch.from_numpy(word.view(np.int32)).cuda()


def empty_like_storage(fmt, h, w, n=None):
    shape = ((n,) if n else ()) + ((h, w) if fmt == "r10" else (h, w, 4))
    return torch.zeros(shape, dtype={"f16": torch.float16, "u8": torch.uint8, "r10": torch.int32}[fmt], device="cuda")


def raw(t):
    return t.view(torch.int16) if t.dtype == torch.float16 else t


def assert_same_bits(a, b, what):
    bad = raw(a) != raw(b)
    n = int(bad.sum())
    assert n == 0, "%s: %d of %d stored values differ (first at %s)" % (what, n, bad.numel(), torch.nonzero(bad)[:3].tolist())


@pytest.mark.parametrize("fmt", ["f16", "u8", "r10"])
@pytest.mark.parametrize("name", list(SHAPES))
def test_strict_easu_is_bit_identical_to_exact(fsr, name, fmt):
    iw, ih, ow, oh = SHAPES[name]
    if fmt != "f16" and name in ("4k_to_8k", "1270p_to_4k", "1662p_to_4k"):
        pytest.skip("UNORM storage: covered by the other shapes")
    src = as_storage(frames.synthetic_frame(iw, ih, k=11), fmt)
    con = fsr.FsrEasuCon(iw, ih, iw, ih, ow, oh)
    ex, st = empty_like_storage(fmt, oh, ow), empty_like_storage(fmt, oh, ow)
    for extra in (0, fsr.FLAG_FRAMES_OVERLAP, fsr.FLAG_NO_FAST_PATHS, fsr.FLAG_OUTPUT_STREAMING):
        ex.zero_(); st.fill_(1)
        fsr.easu(src, ex, con=con, flags=fsr.FLAG_MATH_EXACT | extra)
        fsr.easu(src, st, con=con, flags=fsr.FLAG_MATH_STRICT | extra)
        assert_same_bits(st, ex, "%s %s EASU strict vs EXACT (extra flags 0x%x)" % (name, fmt, extra))


def test_strict_easu_against_the_reference_itself(fsr, ref):
    """Not only through EXACT: the strict image against oracle/_ref's FsrEasuF, on natural content (GUI text, gradients, foliage)."""
    import image_parity
    from test_gpu_fullframe import gpu_assert_exact16
    img = image_parity.natural_frame()
    for ow, oh in ((2954, 1662), (1920, 1080)):
        con = ref.FsrEasuCon(1477, 831, 1477, 831, ow, oh)
        want = ref.easu_f(img.astype(np.float32), ow, oh, con)
        out = torch.zeros(oh, ow, 4, dtype=torch.float16, device="cuda")
        fsr.easu(torch.from_numpy(img).cuda(), out, con=con, flags=fsr.FLAG_MATH_STRICT)
        gpu_assert_exact16(out, want, "natural 1477x831 -> %dx%d EASU strict vs FsrEasuF" % (ow, oh))


@pytest.mark.parametrize("kind", ["adversarial", "hdr_noise", "dark_noise", "primaries", "zeros", "negative"])
@pytest.mark.parametrize("scale", ["2x", "1p5x"])
def test_strict_on_hostile_content(fsr, kind, scale):
    """Content that fails the rounding-boundary test almost everywhere (windows mixing magnitudes: the queue overflows and the
    re-evaluation runs in rounds), that passes it everywhere (flat primaries: constant channels), and values outside {0..1}."""
    iw, ih = 322, 187
    ow, oh = (644, 374) if scale == "2x" else (483, 281)
    g = np.random.default_rng(5)
    if kind == "adversarial":
        img = frames.adversarial_frame(iw, ih, k=3)
    elif kind == "hdr_noise":
        img = np.minimum(np.exp(g.normal(0, 3, (ih, iw, 4))), 60000.0).astype(np.float16)
    elif kind == "dark_noise":
        img = (g.random((ih, iw, 4)) ** 6).astype(np.float16)
    elif kind == "primaries":
        img = np.zeros((ih, iw, 4), np.float16)
        img[:, : iw // 3, 0] = 1.0
        img[:, iw // 3: 2 * iw // 3, 1] = 1.0
        img[:, 2 * iw // 3:, 2] = 0.5
        img[ih // 2:, :, :3] *= np.float16(0.25)
    elif kind == "zeros":
        img = np.zeros((ih, iw, 4), np.float16)
    else:
        img = (g.random((ih, iw, 4)) * 2.0 - 1.0).astype(np.float16)
    img[..., 3] = 1.0
    src = torch.from_numpy(np.ascontiguousarray(img)).cuda()
    con = fsr.FsrEasuCon(iw, ih, iw, ih, ow, oh)
    ex, st = (torch.zeros(oh, ow, 4, dtype=torch.float16, device="cuda") for _ in range(2))
    fsr.easu(src, ex, con=con, flags=fsr.FLAG_MATH_EXACT)
    fsr.easu(src, st, con=con, flags=fsr.FLAG_MATH_STRICT)
    assert_same_bits(st, ex, "%s %s EASU strict vs EXACT" % (kind, scale))
    # the fused launch on the same content: easu(EXACT) -> rcas(default)
    rc = fsr.FsrRcasCon(0.0)
    want, got = torch.zeros_like(ex), torch.zeros_like(ex)
    fsr.rcas(ex, want, con=rc)
    fsr.easu_rcas_fused(src, got, easu_con=con, rcas_con=rc, flags=fsr.FLAG_MATH_STRICT)
    nan = torch.isnan(want) & torch.isnan(got)
    bad = (raw(want) != raw(got)) & ~nan
    assert int(bad.sum()) == 0, "%s %s fused strict vs easu(EXACT) -> rcas(default): %d values differ" % (kind, scale, int(bad.sum()))


def _worst_tiles():
    import json
    import os
    doc = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "strict_worst_tiles.json")))
    return doc["threshold"], doc["tiles"]


@pytest.mark.parametrize("k", range(10))
def test_strict_on_the_adversarial_search_worst_tiles(fsr, k):
    """The input tiles on which an evolution strategy found the default arithmetic furthest from the reference's operation order
    (tools/experiments_r06/strict_adversarial.py: one per ratio of the first search, the two beyond d = 35 of the second; tests/golden/gen_strict_worst_tiles.py), tiled over an image of the
    search's size at the ratio they were found at (every tile position, the recorded one among them: off 2x the sub-texel position is
    rounded from the absolute coordinate): (a) easu(STRICT) stores EXACT's bits; (b) the distance the threshold is a bound of,
    d = |default - EXACT| / (2^-24 M), M = the largest |R|,|G|,|B| of the pixel's 12 taps, stays below the threshold (56); (c) the
    recorded distance is reproduced — the fixture still is the hard case it was when it was recorded."""
    threshold, tiles = _worst_tiles()
    t = tiles[k]
    T, num, den = t["T"], t["num"], t["den"]
    iw, ih = t["in"]
    tile = torch.from_numpy(np.asarray(t["rgb_bits"], np.uint16).reshape(T, T, 3).view(np.int16).copy()).cuda().view(torch.float16)
    src = torch.ones(ih, iw, 4, dtype=torch.float16, device="cuda")
    src[..., :3] = tile.repeat(ih // T, iw // T, 1)
    ow, oh = iw * num // den, ih * num // den
    con = fsr.FsrEasuCon(iw, ih, iw, ih, ow, oh)
    ex, st = (torch.zeros(oh, ow, 4, dtype=torch.float16, device="cuda") for _ in range(2))
    fsr.easu(src, ex, con=con, flags=fsr.FLAG_MATH_EXACT)
    fsr.easu(src, st, con=con, flags=fsr.FLAG_MATH_STRICT)
    assert_same_bits(st, ex, "worst tile %d/%d EASU strict vs EXACT" % (num, den))
    # the bound itself, on binary32 storage (no conversion between the two arithmetics' results and the comparison)
    s32 = src.float().contiguous()
    d32, e32 = (torch.zeros(oh, ow, 4, dtype=torch.float32, device="cuda") for _ in range(2))
    fsr.easu(s32, d32, con=con)
    fsr.easu(s32, e32, con=con, flags=fsr.FLAG_MATH_EXACT)
    delta = (d32[..., :3].double() - e32[..., :3].double()).abs().amax(dim=-1)
    c = np.asarray(con, np.uint32).view(np.float32)
    fx = torch.floor(torch.arange(ow, device="cuda", dtype=torch.float32) * float(c[0]) + float(c[2])).long()  # ffx_fsr1.h:324-326
    fy = torch.floor(torch.arange(oh, device="cuda", dtype=torch.float32) * float(c[1]) + float(c[3])).long()
    mag = s32[..., :3].abs().amax(dim=-1)
    M = torch.zeros(oh, ow, device="cuda")
    for dy, dxs in ((-1, (0, 1)), (0, (-1, 0, 1, 2)), (1, (-1, 0, 1, 2)), (2, (0, 1))):  # the 12 taps b c / e f g h / i j k l / n o
        rows = mag[(fy + dy).clamp(0, ih - 1)]
        for dx in dxs:
            M = torch.maximum(M, rows[:, (fx + dx).clamp(0, iw - 1)])
    d = delta / (M.double().clamp_min(2.0 ** -126) * 2.0 ** -24)
    assert float(d.max()) < threshold, "d = %.2f reaches the F-strict threshold %d" % (float(d.max()), threshold)
    # the recorded position: pixels whose 'f' texel lies in tile `at_tile`
    tx, ty = t["at_tile"]
    own = ((fy // T == ty)[:, None] & (fx // T == tx)[None, :])
    here = float(d[own].max())
    assert here > 0.95 * t["max_d_measured"], "the fixture lost its worst case: d = %.2f at the recorded tile, recorded %.2f" % (here, t["max_d_measured"])


@pytest.mark.parametrize("name", ["540p_to_1080p", "1080p_to_4k", "1440p_to_4k", "831p_to_1080p", "ragged_2x", "ragged_1p7x", "tiny_2x", "minify"])
@pytest.mark.parametrize("fmt", ["f16", "u8"])
def test_every_strict_pipeline_is_easu_exact_then_rcas_default(fsr, name, fmt):
    iw, ih, ow, oh = SHAPES[name]
    src = as_storage(frames.synthetic_frame(iw, ih, k=4), fmt)
    con = fsr.FsrEasuCon(iw, ih, iw, ih, ow, oh)
    rc = fsr.FsrRcasCon(0.25)
    mid, want = empty_like_storage(fmt, oh, ow), empty_like_storage(fmt, oh, ow)
    fsr.easu(src, mid, con=con, flags=fsr.FLAG_MATH_EXACT)
    fsr.rcas(mid, want, con=rc)
    S = fsr.FLAG_MATH_STRICT
    # two dispatches
    mid2, out = empty_like_storage(fmt, oh, ow), empty_like_storage(fmt, oh, ow)
    fsr.easu(src, mid2, con=con, flags=S)
    fsr.rcas(mid2, out, con=rc, flags=S)
    assert_same_bits(mid2, mid, name + " strict intermediary")
    assert_same_bits(out, want, name + " strict two dispatches")
    # the fused launch, alone and with the overlap hint (the walking form at exactly 2x)
    for extra in (0, fsr.FLAG_FRAMES_OVERLAP, fsr.FLAG_NO_FAST_PATHS):
        out.fill_(1)
        fsr.easu_rcas_fused(src, out, easu_con=con, rcas_con=rc, flags=S | extra)
        assert_same_bits(out, want, "%s strict fused launch (extra 0x%x)" % (name, extra))
    # RCAS options travel
    opt = fsr.FLAG_RCAS_DENOISE
    fsr.rcas(mid, want, con=rc, flags=opt)
    out.fill_(1)
    fsr.easu_rcas_fused(src, out, easu_con=con, rcas_con=rc, flags=S | opt)
    assert_same_bits(out, want, name + " strict fused launch with RCAS_DENOISE")
    if fmt != "f16":
        return
    # fsr1_upscale (auto) and a 3-stream pipeline
    fsr.rcas(mid, want, con=rc)
    pipe = fsr.Pipeline(3)
    outs = [torch.zeros_like(want) for _ in range(4)]
    for i, o in enumerate(outs):
        pipe.upscale(src, o, sharpness=0.25, use_rcas=True, fused=(0, 1, 2, 2)[i], flags=S)
    pipe.synchronize()
    pipe.close()
    for i, o in enumerate(outs):
        assert_same_bits(o, want, "%s strict pipeline submission %d" % (name, i))


@pytest.mark.parametrize("steps", [1, 2, 3, 5])
def test_strict_fused_exact_2x_launch_shapes(fsr, steps):
    """Forced run lengths and the tall tile of the exact-2x fused launch (libfsr1_hip_test.so): every shape the same image."""
    lib = fsr._lib.load_test()
    saved = fsr._lib._lib
    iw, ih, ow, oh = 640, 360, 1280, 720
    src = torch.from_numpy(frames.synthetic_frame(iw, ih, k=9)).cuda()
    con = fsr.FsrEasuCon(iw, ih, iw, ih, ow, oh)
    rc = fsr.FsrRcasCon(0.5)
    mid, want, out = (torch.zeros(oh, ow, 4, dtype=torch.float16, device="cuda") for _ in range(3))
    fsr.easu(src, mid, con=con, flags=fsr.FLAG_MATH_EXACT)
    fsr.rcas(mid, want, con=rc)
    try:
        fsr._lib._lib = lib
        for tall in ((0, 1) if steps == 1 else (0,)):
            lib.fsr1_debug_fused_run_steps(steps)
            lib.fsr1_debug_fused_tall_tiles(tall)
            out.fill_(1)
            fsr.easu_rcas_fused(src, out, easu_con=con, rcas_con=rc, flags=fsr.FLAG_MATH_STRICT)
            assert_same_bits(out, want, "strict fused launch, %d steps, tall %d" % (steps, tall))
        # and the stand-alone EASU's two tile heights
        for tall in (0, 1):
            lib.fsr1_debug_easu_tall_tiles(tall)
            out.fill_(1)
            fsr.easu(src, out, con=con, flags=fsr.FLAG_MATH_STRICT)
            assert_same_bits(out, mid, "strict EASU, tall tiles %d" % tall)
    finally:
        lib.fsr1_debug_fused_run_steps(0)
        lib.fsr1_debug_fused_tall_tiles(-1)
        lib.fsr1_debug_easu_tall_tiles(-1)
        fsr._lib._lib = saved


def test_strict_batches_and_bands(fsr):
    iw, ih, ow, oh, n = 480, 270, 960, 540, 3
    batch = np.stack([frames.synthetic_frame(iw, ih, k=k) for k in range(n)])
    src = torch.from_numpy(batch).cuda()
    for (w2, h2) in ((ow, oh), (720, 405)):
        con = fsr.FsrEasuCon(iw, ih, iw, ih, w2, h2)
        ex, st = (torch.zeros(n, h2, w2, 4, dtype=torch.float16, device="cuda") for _ in range(2))
        fsr.easu(src, ex, con=con, flags=fsr.FLAG_MATH_EXACT)
        fsr.easu(src, st, con=con, flags=fsr.FLAG_MATH_STRICT)
        assert_same_bits(st, ex, "strict EASU on a %d-frame batch -> %dx%d" % (n, w2, h2))
        # a band of frame 0: rows [y0, y0 + bh) of the full output
        y0, bh = (h2 // 3) & ~1, h2 // 4
        band = torch.zeros(bh, w2, 4, dtype=torch.float16, device="cuda")
        fsr.easu_band(src[0], band, con, origin=(0, y0), flags=fsr.FLAG_MATH_STRICT)
        assert_same_bits(band, ex[0, y0:y0 + bh], "strict EASU band at row %d -> %dx%d" % (y0, w2, h2))


def test_strict_without_a_strict_variant_takes_exact(fsr):
    """RGBA32F storage has no conversion to test against, `c *= c` and the colour stages have no strict kernels: EXACT's bits."""
    iw, ih, ow, oh = 320, 180, 640, 360
    img = frames.synthetic_frame(iw, ih, k=2, dtype=np.float32)
    src = torch.from_numpy(img).cuda()
    con = fsr.FsrEasuCon(iw, ih, iw, ih, ow, oh)
    rc = fsr.FsrRcasCon(0.25)
    a, b = (torch.zeros(oh, ow, 4, dtype=torch.float32, device="cuda") for _ in range(2))
    fsr.easu(src, a, con=con, flags=fsr.FLAG_MATH_EXACT)
    fsr.easu(src, b, con=con, flags=fsr.FLAG_MATH_STRICT)
    assert torch.equal(a.view(torch.int32), b.view(torch.int32)), "RGBA32F EASU: strict is not EXACT"
    fsr.easu_rcas_fused(src, a, easu_con=con, rcas_con=rc, flags=fsr.FLAG_MATH_EXACT)
    fsr.easu_rcas_fused(src, b, easu_con=con, rcas_con=rc, flags=fsr.FLAG_MATH_STRICT)
    assert torch.equal(a.view(torch.int32), b.view(torch.int32)), "RGBA32F fused launch: strict is not EXACT"
    mid = torch.zeros_like(a)
    fsr.easu(src, mid, con=con, flags=fsr.FLAG_MATH_STRICT)
    fsr.rcas(mid, b, con=rc, flags=fsr.FLAG_MATH_STRICT)
    assert torch.equal(a.view(torch.int32), b.view(torch.int32)), "RGBA32F two dispatches under strict differ from the fused launch"
    s16 = torch.from_numpy(frames.synthetic_frame(iw, ih, k=2)).cuda()
    c, d = (torch.zeros(oh, ow, 4, dtype=torch.float16, device="cuda") for _ in range(2))
    fsr.easu(s16, c, con=con, flags=fsr.FLAG_MATH_EXACT | fsr.FLAG_HDR_SQUARE)
    fsr.easu(s16, d, con=con, flags=fsr.FLAG_MATH_STRICT | fsr.FLAG_HDR_SQUARE)
    assert_same_bits(d, c, "EASU with c *= c under strict")


def test_strict_is_exclusive_with_the_other_arithmetics(fsr):
    src = torch.zeros(8, 8, 4, dtype=torch.float16, device="cuda")
    dst = torch.zeros(16, 16, 4, dtype=torch.float16, device="cuda")
    for other in (fsr.FLAG_MATH_EXACT, fsr.FLAG_MATH_PACKED_FP16):
        with pytest.raises(fsr.Fsr1Error, match="exclusive"):
            fsr.easu(src, dst, flags=fsr.FLAG_MATH_STRICT | other)
