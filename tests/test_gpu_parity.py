"""Parity of the HIP path (through the C ABI) against the CPU oracle, on a real MI355X.

Parity classes (DESIGN.md "Numerics"):
  EXACT   FSR1_FLAG_MATH_EXACT: bit-identical to the CPU-evaluated FsrEasuF / FsrRcasF — compared as raw
          fp32 bits for RGBA32F images, and as binary16 bits (oracle rounded RTNE) for RGBA16F.
  F       default arithmetic: every value within 1 binary16 ULP of the oracle (tolerance stated by
          BASELINE.json's north_star), and at least 99.5 % of the values bit-equal.
"""
import importlib

import numpy as np
import pytest

from conftest import PIXEL_CASES, load_golden

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
frames = importlib.import_module("fidelityfx-fsr_amd.frames")

ULP_TOL = 1            # binary16 ULPs, north_star: "within 1 ULP fp16"
MIN_EXACT_FRACTION = 0.995


def dev(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a)).cuda()
    return t if dtype is None else t.to(dtype)


def host(t):
    torch.cuda.synchronize()
    return t.detach().cpu().numpy()


def h16bits(a):
    return np.ascontiguousarray(a).astype(np.float16).view(np.uint16)


def assert_exact16(gpu_f16, oracle_f32, what=""):
    g, o = h16bits(gpu_f16), h16bits(oracle_f32)
    bad = g != o
    assert not bad.any(), "%s: %d of %d binary16 values differ (first at %s)" % (what, bad.sum(), bad.size, np.argwhere(bad)[:3].tolist())


def assert_exact32(gpu_f32, oracle_f32, what=""):
    g = np.ascontiguousarray(gpu_f32, np.float32).view(np.uint32)
    o = np.ascontiguousarray(oracle_f32, np.float32).view(np.uint32)
    bad = (g != o) & ~(np.isnan(gpu_f32) & np.isnan(oracle_f32))
    assert not bad.any(), "%s: %d of %d fp32 values differ" % (what, bad.sum(), bad.size)


def assert_f_class(gpu, oracle_f32, what=""):
    import cpu_oracle
    d = cpu_oracle.half_ulp_diff(np.asarray(gpu, np.float32), oracle_f32)
    assert d.max() <= ULP_TOL, "%s: max %d binary16 ULP (tolerance %d)" % (what, d.max(), ULP_TOL)
    frac = float((d == 0).mean())
    assert frac >= MIN_EXACT_FRACTION, "%s: only %.4f of values bit-equal" % (what, frac)
    assert not np.isnan(np.asarray(gpu, np.float32)).any()


# ------------------------------------------------------------------------------------------------
# golden fixtures (generated from the reference headers compiled verbatim)
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", PIXEL_CASES)
def test_easu_golden(fsr, name):
    g = load_golden(name)
    oh, ow = g["easu_f"].shape[:2]
    src16 = dev(g["input"])
    out16 = torch.zeros(oh, ow, 4, dtype=torch.float16, device="cuda")
    fsr.easu(src16, out16, con=g["con"], flags=fsr.FLAG_MATH_EXACT)
    assert_exact16(host(out16), g["easu_f"], name + " easu EXACT f16")
    out16.zero_()
    fsr.easu(src16, out16, con=g["con"])
    assert_f_class(host(out16), g["easu_f"], name + " easu F f16")
    src32 = src16.float()
    out32 = torch.zeros(oh, ow, 4, dtype=torch.float32, device="cuda")
    fsr.easu(src32, out32, con=g["con"], flags=fsr.FLAG_MATH_EXACT)
    assert_exact32(host(out32), g["easu_f"], name + " easu EXACT f32")
    out32.zero_()
    fsr.easu(src32, out32, con=g["con"])
    assert_f_class(host(out32), g["easu_f"], name + " easu F f32")
    out32.zero_()
    fsr.easu(src32, out32, con=g["con"], flags=fsr.FLAG_MATH_EXACT | fsr.FLAG_HDR_SQUARE)
    assert_exact32(host(out32), g["easu_f_hdr"], name + " easu EXACT f32 hdr")


@pytest.mark.parametrize("name", PIXEL_CASES)
def test_rcas_golden(fsr, name):
    g = load_golden(name)
    mid16 = dev(g["mid"])
    out16 = torch.zeros_like(mid16)
    flagmap = {0: 0, 1: fsr.FLAG_RCAS_DENOISE, 2: fsr.FLAG_RCAS_PASSTHROUGH_ALPHA,
               3: fsr.FLAG_RCAS_DENOISE | fsr.FLAG_RCAS_PASSTHROUGH_ALPHA}
    for fl, bits in flagmap.items():
        want = g["rcas_f_%d" % fl]
        out16.zero_()
        fsr.rcas(mid16, out16, con=g["rcas_con"], flags=bits | fsr.FLAG_MATH_EXACT)
        assert_exact16(host(out16), want, "%s rcas EXACT f16 flags %d" % (name, fl))
        out16.zero_()
        fsr.rcas(mid16, out16, con=g["rcas_con"], flags=bits)
        assert_f_class(host(out16), want, "%s rcas F f16 flags %d" % (name, fl))
        mid32 = mid16.float()
        out32 = torch.zeros_like(mid32)
        fsr.rcas(mid32, out32, con=g["rcas_con"], flags=bits | fsr.FLAG_MATH_EXACT)
        assert_exact32(host(out32), want, "%s rcas EXACT f32 flags %d" % (name, fl))
    out32.zero_()
    fsr.rcas(mid16.float(), out32, con=g["rcas_con"], flags=fsr.FLAG_HDR_SQUARE | fsr.FLAG_MATH_EXACT)
    assert_exact32(host(out32), g["rcas_f_hdr"], name + " rcas EXACT f32 hdr")


def test_survey_kat_b2_on_gpu(fsr):
    """SURVEY.md Appendix B.2 frame, fp32 I/O, intermediate not rounded: bit-exact chain."""
    g = load_golden("kat_b2")
    src = dev(g["input"])
    mid = torch.zeros(72, 128, 4, dtype=torch.float32, device="cuda")
    out = torch.zeros_like(mid)
    fsr.easu(src, mid, con=g["con"], flags=fsr.FLAG_MATH_EXACT)
    fsr.rcas(mid, out, sharpness=0.25, flags=fsr.FLAG_MATH_EXACT)
    assert_exact32(host(mid), g["easu_f"], "B.2 easu")
    assert_exact32(host(out), g["rcas_f"], "B.2 rcas")
    assert abs(host(out).astype(np.float64).sum() - 19292.897310251) < 1e-6


# ------------------------------------------------------------------------------------------------
# live oracle on larger / awkward shapes
# ------------------------------------------------------------------------------------------------
SHAPES = [
    (480, 270, 960, 540),     # 2.0x
    (369, 208, 480, 270),     # 1.3x true-ratio preset shape scaled down
    (564, 317, 960, 540),     # 1.7x
    (640, 360, 960, 540),     # 1.5x
    (97, 61, 131, 83),        # ragged, not tile aligned
    (5, 3, 17, 9),            # smaller than one tile, tiny input
    (64, 16, 64, 16),         # ratio 1.0
    (300, 200, 200, 133),     # mild minification (footprint larger than the tile)
]


@pytest.mark.parametrize("shape", SHAPES, ids=lambda s: "%dx%d_to_%dx%d" % s)
def test_easu_vs_oracle(fsr, port, shape):
    iw, ih, ow, oh = shape
    img = frames.synthetic_frame(iw, ih, k=1, dtype=np.float16)
    con = fsr.FsrEasuCon(iw, ih, iw, ih, ow, oh)
    want = port.easu_f(img.astype(np.float32), ow, oh, con)
    out = torch.zeros(oh, ow, 4, dtype=torch.float16, device="cuda")
    fsr.easu(dev(img), out, con=con, flags=fsr.FLAG_MATH_EXACT)
    assert_exact16(host(out), want, "easu EXACT")
    out.zero_()
    fsr.easu(dev(img), out, con=con)
    assert_f_class(host(out), want, "easu F")


@pytest.mark.parametrize("shape", [(960, 540), (131, 83), (64, 16), (65, 17), (3, 2), (1, 1)], ids=lambda s: "%dx%d" % s)
def test_rcas_vs_oracle(fsr, port, shape):
    w, h = shape
    img = frames.synthetic_frame(w, h, k=2, dtype=np.float16)
    for stops in (0.0, 0.25, 1.5):
        con = fsr.FsrRcasCon(stops)
        want = port.rcas_f(img.astype(np.float32), con)
        out = torch.zeros(h, w, 4, dtype=torch.float16, device="cuda")
        fsr.rcas(dev(img), out, con=con, flags=fsr.FLAG_MATH_EXACT)
        assert_exact16(host(out), want, "rcas EXACT %g" % stops)
        out.zero_()
        fsr.rcas(dev(img), out, con=con)
        assert_f_class(host(out), want, "rcas F %g" % stops)


def test_rcas_black_white_primaries(fsr, port):
    """0*inf NaNs inside RCAS must be dropped by max(); exact 0.0 / 1.0 regions."""
    img = np.zeros((40, 72, 4), np.float16)
    img[..., 3] = 1
    img[:, 8:16, :3] = 1.0
    img[:, 16:24, 0] = 1.0
    img[:, 24:32, 1] = 1.0
    img[:, 32:40, 2] = 1.0
    img[::2, 40:48, :3] = 1.0
    img[:, 48:56, 1] = 0.5
    img[17, 52, 1] = 0.8
    con = fsr.FsrRcasCon(0.0)
    want = port.rcas_f(img.astype(np.float32), con)
    assert not np.isnan(want).any()
    out = torch.zeros(40, 72, 4, dtype=torch.float16, device="cuda")
    for fl in (fsr.FLAG_MATH_EXACT, 0):
        out.fill_(7)
        fsr.rcas(dev(img), out, con=con, flags=fl)
        got = host(out)
        assert not np.isnan(got.astype(np.float32)).any()
        if fl:
            assert_exact16(got, want, "rcas specials EXACT")
        else:
            assert_f_class(got, want, "rcas specials F")


def test_easu_dynamic_resolution(fsr, port):
    """FsrEasuConOffset: viewport (40x30 at offset 8,6) inside a 64x48 resource; taps clamp at the resource edge."""
    img = frames.synthetic_frame(64, 48, k=2, dtype=np.float16)
    con = fsr.FsrEasuConOffset(40, 30, 64, 48, 80, 60, 8, 6)
    want = port.easu_f(img.astype(np.float32), 80, 60, con)
    out = torch.zeros(60, 80, 4, dtype=torch.float16, device="cuda")
    fsr.easu(dev(img), out, con=con, flags=fsr.FLAG_MATH_EXACT)
    assert_exact16(host(out), want, "easu offset EXACT")


def test_batch_and_strides(fsr, port):
    """Frames of one launch are independent; row pitch and frame stride are honoured."""
    iw, ih, ow, oh, n = 96, 54, 192, 108, 5
    batch = np.stack([frames.synthetic_frame(iw, ih, k=k, dtype=np.float16) for k in range(n)])
    con = fsr.FsrEasuCon(iw, ih, iw, ih, ow, oh)
    padded_in = torch.zeros(n, ih + 3, iw + 5, 4, dtype=torch.float16, device="cuda")
    padded_in[:, :ih, :iw] = dev(batch)
    padded_out = torch.full((n, oh + 2, ow + 7, 4), 9.0, dtype=torch.float16, device="cuda")
    fsr.easu(padded_in[:, :ih, :iw], padded_out[:, :oh, :ow], con=con, flags=fsr.FLAG_MATH_EXACT)
    got = host(padded_out)
    for k in range(n):
        want = port.easu_f(batch[k].astype(np.float32), ow, oh, con)
        assert_exact16(got[k, :oh, :ow], want, "frame %d" % k)
    assert np.all(got[:, oh:, :, :] == 9.0) and np.all(got[:, :, ow:, :] == 9.0), "wrote outside the output view"
    # RCAS on the same padded layout
    sharp = torch.full_like(padded_out, 5.0)
    rc = fsr.FsrRcasCon(0.25)
    fsr.rcas(padded_out[:, :oh, :ow], sharp[:, :oh, :ow], con=rc, flags=fsr.FLAG_MATH_EXACT)
    got2 = host(sharp)
    for k in range(n):
        want = port.rcas_f(got[k, :oh, :ow].astype(np.float32), rc)
        assert_exact16(got2[k, :oh, :ow], want, "rcas frame %d" % k)
    assert np.all(got2[:, oh:, :, :] == 5.0) and np.all(got2[:, :, ow:, :] == 5.0)


def test_upscale_and_filter_mirror(fsr, port):
    """fsr1_upscale / FSR_Filter.Upscale == FsrEasuCon + EASU + FsrRcasCon + RCAS (FSR_Filter.cpp:101-141)."""
    iw, ih, ow, oh = 160, 90, 320, 180
    img = frames.synthetic_frame(iw, ih, k=4, dtype=np.float16)
    src = dev(img)
    dst = torch.zeros(oh, ow, 4, dtype=torch.float16, device="cuda")
    filt = fsr.FSR_Filter()
    filt.OnCreate(slowFallback=True, exact=True)
    filt.OnCreateWindowSizeDependentResources(src, dst, ow, oh, hdr=False)
    state = fsr.State(iw, ih, bUseRcas=True, rcasAttenuation=0.25)
    filt.Upscale(ow, oh, state)
    con = port.FsrEasuCon(iw, ih, iw, ih, ow, oh)
    mid = port.easu_f(img.astype(np.float32), ow, oh, con).astype(np.float16).astype(np.float32)
    want = port.rcas_f(mid, port.FsrRcasCon(0.25))
    assert_exact16(host(dst), want, "Upscale easu+rcas")
    assert_exact16(host(filt.m_intermediary), mid, "intermediary")
    # EASU only, hdr -> Sample.x = 1 on EASU (FSR_Filter.cpp:107)
    state.bUseRcas = False
    filt.Upscale(ow, oh, state, hdr=True)
    want = port.easu_f(img.astype(np.float32), ow, oh, con, 4)
    assert_exact16(host(dst), want, "Upscale easu-only hdr")
    # hdr with RCAS: square applied after RCAS only (:125)
    state.bUseRcas = True
    filt.Upscale(ow, oh, state, hdr=True)
    want = port.rcas_f(mid, port.FsrRcasCon(0.25), 4)
    assert_exact16(host(dst), want, "Upscale easu+rcas hdr")
    filt.OnDestroy()


@pytest.mark.parametrize("shape", [(480, 270, 960, 540), (97, 61, 131, 83), (640, 360, 960, 540), (5, 3, 17, 9), (64, 16, 64, 16)],
                         ids=lambda s: "%dx%d_to_%dx%d" % s)
def test_fused_equals_two_pass(fsr, port, shape):
    """BASELINE config 4: the LDS-fused EASU->RCAS launch is bit-identical to the two dispatches with an
    intermediary of the output's format, in both arithmetic modes, with every RCAS option; and (EXACT)
    to the CPU oracle chain FsrEasuF -> binary16 -> FsrRcasF."""
    iw, ih, ow, oh = shape
    img = frames.synthetic_frame(iw, ih, k=6, dtype=np.float16)
    src = dev(img)
    con = fsr.FsrEasuCon(iw, ih, iw, ih, ow, oh)
    rc = fsr.FsrRcasCon(0.25)
    for dtype in (torch.float16, torch.float32):
        s = src.to(dtype)
        for math in (fsr.FLAG_MATH_EXACT, 0):
            for opts in (0, fsr.FLAG_RCAS_DENOISE, fsr.FLAG_RCAS_PASSTHROUGH_ALPHA | fsr.FLAG_HDR_SQUARE):
                mid = torch.zeros(oh, ow, 4, dtype=dtype, device="cuda")
                two = torch.zeros_like(mid)
                one = torch.full_like(mid, 3.0)
                fsr.easu(s, mid, con=con, flags=math)
                fsr.rcas(mid, two, con=rc, flags=math | opts)
                fsr.easu_rcas_fused(s, one, easu_con=con, rcas_con=rc, flags=math | opts)
                torch.cuda.synchronize()
                assert torch.equal(one.view(torch.int16 if dtype == torch.float16 else torch.int32),
                                   two.view(torch.int16 if dtype == torch.float16 else torch.int32)), (dtype, math, opts)
    want_mid = port.easu_f(img.astype(np.float32), ow, oh, con).astype(np.float16).astype(np.float32)
    want = port.rcas_f(want_mid, rc)
    out = torch.zeros(oh, ow, 4, dtype=torch.float16, device="cuda")
    fsr.easu_rcas_fused(src, out, easu_con=con, rcas_con=rc, flags=fsr.FLAG_MATH_EXACT)
    assert_exact16(host(out), want, "fused EXACT vs oracle chain")


def test_fused_through_upscale(fsr, port):
    iw, ih, ow, oh = 160, 90, 320, 180
    img = frames.synthetic_frame(iw, ih, k=4, dtype=np.float16)
    src = dev(img)
    dst = torch.zeros(oh, ow, 4, dtype=torch.float16, device="cuda")
    filt = fsr.FSR_Filter()
    filt.OnCreate(slowFallback=True, exact=True, fused=True)
    filt.OnCreateWindowSizeDependentResources(src, dst, ow, oh)
    filt.Upscale(ow, oh, fsr.State(iw, ih, bUseRcas=True, rcasAttenuation=0.25))
    con = port.FsrEasuCon(iw, ih, iw, ih, ow, oh)
    mid = port.easu_f(img.astype(np.float32), ow, oh, con).astype(np.float16).astype(np.float32)
    assert_exact16(host(dst), port.rcas_f(mid, port.FsrRcasCon(0.25)), "fused Upscale")
    assert filt.m_intermediary is None


# ------------------------------------------------------------------------------------------------
# BASELINE.json full sizes: oracle on row bands + size-independent properties
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("shape", [(1920, 1080, 3840, 2160), (2560, 1440, 3840, 2160), (3840, 2160, 7680, 4320), (960, 540, 1920, 1080)],
                         ids=["1080p_to_4k", "1440p_to_4k", "4k_to_8k", "540p_to_1080p"])
def test_full_size_bands_and_properties(fsr, port, shape):
    iw, ih, ow, oh = shape
    img = frames.synthetic_frame(iw, ih, k=0, dtype=np.float16)
    src = dev(img)
    con = fsr.FsrEasuCon(iw, ih, iw, ih, ow, oh)
    rc = fsr.FsrRcasCon(0.25)
    mid = torch.zeros(oh, ow, 4, dtype=torch.float16, device="cuda")
    out = torch.zeros_like(mid)
    fsr.easu(src, mid, con=con)
    fsr.rcas(mid, out, con=rc)
    got_mid, got_out = host(mid), host(out)
    img32 = img.astype(np.float32)
    mid32 = got_mid.astype(np.float32)
    for y0 in (0, oh // 2 - 8, oh - 16):  # top edge, middle, bottom edge
        rows = (y0, y0 + 16)
        want = port.easu_f(img32, ow, oh, con, 0, rows)[y0:y0 + 16]
        assert_f_class(got_mid[y0:y0 + 16], want, "easu rows %d" % y0)
        want = port.rcas_f(mid32, rc, 0, rows)[y0:y0 + 16]   # RCAS judged on identical input (the GPU's own EASU output)
        assert_f_class(got_out[y0:y0 + 16], want, "rcas rows %d" % y0)
    # properties: output alpha is 1, values stay inside the input's range per channel (EASU derings to the
    # 2x2 neighbourhood: ffx_fsr1.h:437), a second run is bit-identical (determinism)
    assert np.all(got_mid[..., 3] == 1.0) and np.all(got_out[..., 3] == 1.0)
    for c in range(3):
        assert got_mid[..., c].min() >= img[..., c].min() and got_mid[..., c].max() <= img[..., c].max()
    mid2 = torch.zeros_like(mid)
    fsr.easu(src, mid2, con=con)
    assert torch.equal(mid, mid2)
    # the single-launch pipeline (BASELINE configs[3]) gives the very same image at full size
    fsr.easu_rcas_fused(src, mid2, easu_con=con, rcas_con=rc)
    assert torch.equal(out, mid2)


def test_batch_sharded_like_config_3(fsr, port):
    """BASELINE configs[2]: a batch of 1440p->4K frames, 8 per GPU: one launch over frames = 8 equals 8 single-frame
    launches bit for bit (frames are independent units), and frame k of the batch matches the oracle on a band."""
    iw, ih, ow, oh, n = 2560, 1440, 3840, 2160, 8
    base = [dev(frames.synthetic_frame(iw, ih, k=k, dtype=np.float16)) for k in range(2)]
    src = torch.stack([torch.roll(base[f % 2], shifts=(3 * f, 5 * f), dims=(0, 1)) for f in range(n)]).contiguous()
    con = fsr.FsrEasuCon(iw, ih, iw, ih, ow, oh)
    rc = fsr.FsrRcasCon(0.25)
    batch = torch.zeros(n, oh, ow, 4, dtype=torch.float16, device="cuda")
    fsr.easu_rcas_fused(src, batch, easu_con=con, rcas_con=rc)
    one = torch.zeros(oh, ow, 4, dtype=torch.float16, device="cuda")
    for f in (0, 3, 7):
        fsr.easu_rcas_fused(src[f], one, easu_con=con, rcas_con=rc)
        assert torch.equal(batch[f], one), "frame %d of the batch differs from its single-frame launch" % f
    f, y0 = 5, 1000
    img32 = host(src[f]).astype(np.float32)
    mid = torch.zeros(oh, ow, 4, dtype=torch.float16, device="cuda")
    fsr.easu(src[f], mid, con=con)
    want = port.rcas_f(host(mid).astype(np.float32), rc, 0, (y0, y0 + 8))[y0:y0 + 8]
    assert_f_class(host(batch[f])[y0:y0 + 8], want, "batch frame %d" % f)
    assert_f_class(host(mid)[y0:y0 + 8], port.easu_f(img32, ow, oh, con, 0, (y0, y0 + 8))[y0:y0 + 8], "batch easu frame %d" % f)


def test_constant_image_is_a_fixed_point(fsr):
    """EASU of a constant image returns it (weights normalise, dering clamp = identity); RCAS of a constant
    interior returns (lobe*4c + c) * APrxMedRcpF1(4*lobe+1) = c up to the error of the medium-precision
    reciprocal (one Newton step from the integer estimate, ffx_a.h:1844: below 0.5 % relative)."""
    for val in (0.0, 0.25, 0.7, 1.0):
        src = torch.full((45, 80, 4), val, dtype=torch.float16, device="cuda")
        src[..., 3] = 1
        mid = torch.zeros(90, 160, 4, dtype=torch.float16, device="cuda")
        fsr.easu(src, mid)
        assert torch.equal(mid[..., :3], torch.full_like(mid[..., :3], val))
        out = torch.zeros_like(mid)
        fsr.rcas(mid, out)
        inner = host(out)[1:-1, 1:-1, :3].astype(np.float32)
        assert np.abs(inner - np.float32(np.float16(val))).max() <= 0.005 * val + 1e-7


def test_dispatches_are_graph_capturable(fsr):
    """the passes are plain asynchronous launches on the caller's stream, so a frame loop can be captured
    into a hipGraph and replayed (launch-bound small frames); replay gives the very same image."""
    iw, ih, ow, oh = 320, 180, 640, 360
    src = dev(frames.synthetic_frame(iw, ih, k=3, dtype=np.float16))
    mid = torch.zeros(oh, ow, 4, dtype=torch.float16, device="cuda")
    eager = torch.zeros_like(mid)
    fus = torch.zeros_like(mid)
    con, rc = fsr.FsrEasuCon(iw, ih, iw, ih, ow, oh), fsr.FsrRcasCon(0.25)
    fsr.easu(src, mid, con=con)
    fsr.rcas(mid, eager, con=rc)
    torch.cuda.synchronize()
    out = torch.zeros_like(mid)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fsr.easu(src, mid, con=con)
        fsr.rcas(mid, out, con=rc)
        fsr.easu_rcas_fused(src, fus, easu_con=con, rcas_con=rc)
    out.zero_()
    fus.zero_()
    mid.zero_()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, eager) and torch.equal(fus, eager)
    # new input contents, same buffers: the graph recomputes
    src.copy_(dev(frames.synthetic_frame(iw, ih, k=4, dtype=np.float16)))
    g.replay()
    fsr.easu(src, mid, con=con)
    fsr.rcas(mid, eager, con=rc)
    torch.cuda.synchronize()
    assert torch.equal(out, eager)


@pytest.mark.parametrize("shape", [(480, 270, 960, 540), (37, 23, 74, 46), (1, 1, 2, 2), (33, 9, 66, 18), (960, 540, 1920, 1080)],
                         ids=lambda s: "%dx%d_to_%dx%d" % s)
@pytest.mark.parametrize("fmt", ["f16", "f32", "rgba8"])
def test_exact_2x_fast_path_is_bit_identical(fsr, shape, fmt):
    """At exactly 2x (con0 = {1/2, 1/2, -1/4, -1/4}) the EASU kernel whose lanes own 2x2 output quads is selected (shifted
    tiles); it must reproduce the generic kernel bit for bit (same per-pixel arithmetic on the same values), for both
    arithmetics, incl. sizes that are / are not multiples of the tile and the 1x1 input."""
    iw, ih, ow, oh = shape
    img = frames.synthetic_frame(iw, ih, k=7, dtype=np.float32)
    if fmt == "f16":
        src, dt = dev(img.astype(np.float16)), torch.float16
    elif fmt == "f32":
        src, dt = dev(img), torch.float32
    else:
        src, dt = dev(np.floor(np.clip(img, 0, 1) * 255.0 + 0.5).astype(np.uint8)), torch.uint8
    for flags in (0, fsr.FLAG_MATH_EXACT):
        fast = torch.zeros(oh, ow, 4, dtype=dt, device="cuda")
        slow = torch.zeros(oh, ow, 4, dtype=dt, device="cuda")
        fsr.easu(src, fast, flags=flags)
        fsr.easu(src, slow, flags=flags | fsr.FLAG_NO_FAST_PATHS)
        assert torch.equal(fast, slow), "easu exact-2x path differs (flags %d)" % flags


def test_upscale_auto_pipeline(fsr, port):
    """fsr1_params.fused = 2 picks the fused launch at exactly 2x (its quad form wins at every size) and, at other ratios, where
    a frame is launch-bound (<= 3 Mpixel of output per launch), the two dispatches above (round-2 measurements,
    include/fsr1_hip.h); the image is the same either way (fused == two-pass bit for bit), so only the choice itself needs
    checking: the intermediary is written iff two-pass ran."""
    for (iw, ih, ow, oh), expect_two_pass in (((160, 90, 320, 180), False), ((160, 90, 240, 135), False), ((1280, 720, 2560, 1440), False),
                                              ((1707, 960, 2560, 1440), True),
                                              # 2x minification: EASU's tile fits a CU's LDS, the fused tile does not — auto keeps
                                              # the two dispatches it was given an intermediary for instead of failing
                                              ((640, 360, 320, 180), True)):
        src = dev(frames.synthetic_frame(iw, ih, k=2, dtype=np.float16))
        dst = torch.zeros(oh, ow, 4, dtype=torch.float16, device="cuda")
        filt = fsr.FSR_Filter()
        filt.OnCreate(slowFallback=True, exact=True, fused="auto")
        filt.OnCreateWindowSizeDependentResources(src, dst, ow, oh)
        filt.m_intermediary.fill_(-1.0)
        filt.Upscale(ow, oh, fsr.State(iw, ih, bUseRcas=True, rcasAttenuation=0.25))
        touched = bool((host(filt.m_intermediary) != -1.0).any())
        assert touched == expect_two_pass
        con = port.FsrEasuCon(iw, ih, iw, ih, ow, oh)
        mid = port.easu_f(host(src).astype(np.float32), ow, oh, con).astype(np.float16).astype(np.float32)
        assert_exact16(host(dst), port.rcas_f(mid, port.FsrRcasCon(0.25)), "auto pipeline")
        filt.OnDestroy()


def test_exact_2x_fast_path_batch_and_pitches(fsr):
    """The exact-2x EASU variant on a batch with padded rows and frames (views into larger allocations), HDR square on."""
    n, iw, ih = 3, 70, 37
    ow, oh = 2 * iw, 2 * ih
    big_in = torch.zeros(n, ih + 3, iw + 5, 4, dtype=torch.float16, device="cuda")
    src = big_in[:, :ih, :iw]
    for f in range(n):
        src[f].copy_(dev(frames.synthetic_frame(iw, ih, k=10 + f, dtype=np.float16)))
    outs = []
    for flags in (fsr.FLAG_HDR_SQUARE, fsr.FLAG_HDR_SQUARE | fsr.FLAG_NO_FAST_PATHS):
        big_out = torch.full((n, oh + 2, ow + 6, 4), -3.0, dtype=torch.float16, device="cuda")
        dst = big_out[:, :oh, :ow]
        fsr.easu(src, dst, flags=flags)
        torch.cuda.synchronize()
        assert float((big_out[:, oh:] != -3.0).sum()) == 0 and float((big_out[:, :, ow:] != -3.0).sum()) == 0  # padding untouched
        outs.append(dst.clone())
    assert torch.equal(outs[0], outs[1])


def test_random_shapes_sweep(fsr, port):
    """Seeded sweep over random sizes, ratios (0.35x .. 4.3x per axis, also anisotropic), dynamic-resolution viewports
    and offsets, formats and flags, small enough for the CPU oracle: EXACT arithmetic must be bit-identical, the fused
    launch must equal the two dispatches, the default arithmetic stays within 1 binary16 ULP."""
    rng = np.random.default_rng(20260922)
    cases = 0
    for it in range(48):
        iw, ih = int(rng.integers(1, 90)), int(rng.integers(1, 60))
        kind = it % 4
        if kind == 0:    # exact 2x (fast path)
            ow, oh = 2 * iw, 2 * ih
        elif kind == 1:  # up to ~4.3x
            ow, oh = int(iw * rng.uniform(1.0, 4.3)) + 1, int(ih * rng.uniform(1.0, 4.3)) + 1
        elif kind == 2:  # anisotropic / slight minification
            ow, oh = max(1, int(iw * rng.uniform(0.5, 3.0))), max(1, int(ih * rng.uniform(0.35, 2.0)))
        else:            # viewport inside a larger resource + offset
            ow, oh = int(iw * rng.uniform(1.0, 2.5)) + 1, int(ih * rng.uniform(1.0, 2.5)) + 1
        img = frames.synthetic_frame(iw, ih, k=it, dtype=np.float32)
        img[rng.integers(0, ih), rng.integers(0, iw), :3] = 0.0
        img[rng.integers(0, ih), rng.integers(0, iw), :3] = 1.0
        if kind == 3 and iw > 4 and ih > 4:
            vw, vh = int(rng.integers(2, iw)), int(rng.integers(2, ih))
            offx, offy = float(rng.integers(0, iw - vw + 1)), float(rng.integers(0, ih - vh + 1))
            con = port.FsrEasuConOffset(vw, vh, iw, ih, ow, oh, offx, offy)
        else:
            con = port.FsrEasuCon(iw, ih, iw, ih, ow, oh)
        stops = float(rng.choice([0.0, 0.25, 1.0, 2.0]))
        rcon = port.FsrRcasCon(stops)
        rflag = int(rng.choice([0, fsr.FLAG_RCAS_DENOISE, fsr.FLAG_RCAS_PASSTHROUGH_ALPHA, fsr.FLAG_HDR_SQUARE]))
        oflag = {0: 0, fsr.FLAG_RCAS_DENOISE: 1, fsr.FLAG_RCAS_PASSTHROUGH_ALPHA: 2, fsr.FLAG_HDR_SQUARE: 4}[rflag]
        f16 = bool(it % 3)
        src = dev(img.astype(np.float16) if f16 else img)
        dt = torch.float16 if f16 else torch.float32
        img_in = host(src).astype(np.float32)
        want_mid = port.easu_f(img_in, ow, oh, con)
        for exact in (True, False):
            fl = fsr.FLAG_MATH_EXACT if exact else 0
            mid = torch.zeros(oh, ow, 4, dtype=dt, device="cuda")
            out = torch.zeros_like(mid)
            fus = torch.zeros_like(mid)
            fsr.easu(src, mid, con=con, flags=fl)
            fsr.rcas(mid, out, con=rcon, flags=fl | rflag)
            fsr.easu_rcas_fused(src, fus, easu_con=con, rcas_con=rcon, flags=fl | rflag)
            got_mid, got_out = host(mid), host(out)
            what = "case %d %dx%d->%dx%d kind %d %s" % (it, iw, ih, ow, oh, kind, "exact" if exact else "f")
            assert torch.equal(out, fus), what + ": fused != two-pass"
            want_out = port.rcas_f(got_mid.astype(np.float32), rcon, oflag)
            if exact:
                (assert_exact16 if f16 else assert_exact32)(got_mid, want_mid, what + " easu")
                (assert_exact16 if f16 else assert_exact32)(got_out, want_out, what + " rcas")
            else:
                import cpu_oracle
                assert cpu_oracle.half_ulp_diff(got_mid.astype(np.float32), want_mid).max() <= 1, what + " easu"
                assert cpu_oracle.half_ulp_diff(got_out.astype(np.float32), want_out).max() <= 1, what + " rcas"
            cases += 1
    assert cases == 96


def test_output_store_policy_does_not_change_pixels(fsr):
    """FSR1_FLAG_OUTPUT_STREAMING (non-temporal stores; the default of RCAS and the fused launch) and
    FSR1_FLAG_OUTPUT_CACHED (plain stores; EASU's default) store the very same image, for every pass and both shapes."""
    for (iw, ih, ow, oh) in ((160, 90, 320, 180), (97, 61, 131, 83)):
        src = dev(frames.synthetic_frame(iw, ih, k=12, dtype=np.float16))
        outs = []
        for policy in (fsr.FLAG_OUTPUT_CACHED, fsr.FLAG_OUTPUT_STREAMING, 0):
            mid = torch.zeros(oh, ow, 4, dtype=torch.float16, device="cuda")
            out = torch.zeros_like(mid)
            fus = torch.zeros_like(mid)
            fsr.easu(src, mid, flags=policy)
            fsr.rcas(mid, out, flags=policy)
            fsr.easu_rcas_fused(src, fus, flags=policy)
            hmid = torch.zeros_like(mid)
            hout = torch.zeros_like(mid)
            fsr.easu(src, hmid, flags=policy | fsr.FLAG_MATH_PACKED_FP16)
            fsr.rcas(hmid, hout, flags=policy | fsr.FLAG_MATH_PACKED_FP16)
            outs.append((mid, out, fus, hmid, hout))
        for other in outs[1:]:
            for a, b in zip(outs[0], other):
                assert torch.equal(a.view(torch.int16), b.view(torch.int16))


@pytest.mark.parametrize("shape", [(31, 7), (32, 8), (30, 6), (31, 8), (63, 15), (62, 14), (1, 1), (2, 3), (100, 50), (125, 29)], ids=lambda s: "%dx%d" % s)
def test_fused_exact_2x_quad_form(fsr, shape):
    """fsr1_fused_s2.hip (62-pixel columns walked in 16-row steps, a quad of the step's 64 x 16 EASU pixels per lane, DPP-only RCAS neighbours) against the generic
    fused kernel (FSR1_FLAG_NO_FAST_PATHS) and the two dispatches, at sizes on and around its tile boundaries, for every storage
    format, both arithmetics, the RCAS options, batches and padded rows: bit-identical."""
    iw, ih = shape
    ow, oh = 2 * iw, 2 * ih
    n = 3
    base = np.stack([frames.synthetic_frame(iw, ih, k=20 + f, dtype=np.float16) for f in range(n)])
    for dt in (torch.float16, torch.float32, torch.uint8):
        if dt == torch.uint8:
            src_c = (dev(base).float().clamp(0, 1) * 255 + 0.5).floor().to(torch.uint8)
        else:
            src_c = dev(base).to(dt)
        big_in = torch.zeros(n, ih + 2, iw + 3, 4, dtype=dt, device="cuda")
        src = big_in[:, :ih, :iw]
        src.copy_(src_c)
        for flags in (0, fsr.FLAG_MATH_EXACT, fsr.FLAG_RCAS_DENOISE | fsr.FLAG_RCAS_PASSTHROUGH_ALPHA, fsr.FLAG_MATH_EXACT | fsr.FLAG_HDR_SQUARE):
            outs = []
            for extra in (0, fsr.FLAG_NO_FAST_PATHS):
                big_out = torch.full((n, oh + 1, ow + 5, 4), 7, dtype=dt, device="cuda")
                dst = big_out[:, :oh, :ow]
                fsr.easu_rcas_fused(src, dst, sharpness=0.3, flags=flags | extra)
                torch.cuda.synchronize()
                assert bool((big_out[:, oh:] == 7).all()) and bool((big_out[:, :, ow:] == 7).all()), "wrote outside the output view"
                outs.append(dst.clone())
            mid = torch.zeros(n, oh, ow, 4, dtype=dt, device="cuda")
            two = torch.zeros_like(mid)
            fsr.easu(src, mid, flags=flags & (fsr.FLAG_MATH_EXACT))
            fsr.rcas(mid, two, sharpness=0.3, flags=flags)
            raw = {torch.float16: torch.int16, torch.float32: torch.int32, torch.uint8: torch.uint8}[dt]
            assert torch.equal(outs[0].view(raw), outs[1].view(raw)), "quad form != generic fused (%s, flags %d)" % (dt, flags)
            assert torch.equal(outs[0].view(raw), two.view(raw)), "quad form != two dispatches (%s, flags %d)" % (dt, flags)


@pytest.mark.parametrize("steps", [1, 2, 3, 4, 5, 9, 11])
@pytest.mark.parametrize("shape", [(97, 160), (31, 75), (64, 40), (70, 9)], ids=lambda s: "%dx%d" % s)
def test_fused_exact_2x_run_steps(fsr, shape, steps):
    """The exact-2x fused kernel walks down its 62-pixel column in `steps` steps of 16 EASU rows, carrying the last two rows of a
    step to the next in an 18-row LDS ring (fsr1_fused_s2.hip).  The host picks `steps` from the launch's size; whatever it is —
    forced here through the fsr1_debug_fused_run_steps test hook, 9 steps take the ring through every one of its positions — the image is the same,
    bit for bit, as the two dispatches', for whole images, batches and row bands."""
    iw, ih = shape
    ow, oh = 2 * iw, 2 * ih
    n = 2
    src = dev(np.stack([frames.synthetic_frame(iw, ih, k=40 + f, dtype=np.float16) for f in range(n)]))
    for flags in (0, fsr.FLAG_MATH_EXACT | fsr.FLAG_RCAS_DENOISE):
        mid = torch.zeros(n, oh, ow, 4, dtype=torch.float16, device="cuda")
        two = torch.zeros_like(mid)
        fsr.easu(src, mid, flags=flags & fsr.FLAG_MATH_EXACT)
        fsr.rcas(mid, two, sharpness=0.3, flags=flags)
        with fsr._lib.test_hooks() as hooks:  # libfsr1_hip_test.so (include/fsr1_hip_test.h); the two dispatches above ran in the product library
            hooks.fsr1_debug_fused_run_steps(steps)
            big_out = torch.full((n, oh + 1, ow + 5, 4), 7, dtype=torch.float16, device="cuda")
            dst = big_out[:, :oh, :ow]
            fsr.easu_rcas_fused(src, dst, sharpness=0.3, flags=flags)
            torch.cuda.synchronize()
            assert bool((big_out[:, oh:] == 7).all()) and bool((big_out[:, :, ow:] == 7).all()), "wrote outside the output view"
            assert torch.equal(dst.view(torch.int16), two.view(torch.int16)), "steps %d != two dispatches (flags %d)" % (steps, flags)
            band = torch.full_like(two[0], -1.0)
            cuts = [0, (oh // 3) & ~1, (2 * oh // 3) & ~1, oh]
            for y0, y1 in zip(cuts, cuts[1:]):
                if y1 > y0:
                    fsr.upscale_band(src[0], band[y0:y1], (ow, oh), (y0, y1), sharpness=0.3, flags=flags, fused=True)
            assert torch.equal(band.view(torch.int16), two[0].view(torch.int16)), "bands at steps %d != two dispatches (flags %d)" % (steps, flags)


@pytest.mark.parametrize("shape", [(97, 160), (31, 75), (64, 40), (70, 9), (1, 1), (200, 31)], ids=lambda s: "%dx%d" % s)
def test_fused_exact_2x_tall_tiles(fsr, shape):
    """The one-step exact-2x fused launch of a frame that fills the chip runs 512-thread workgroups on 62 x 30 tiles (32 EASU rows per
    step, fsr1_fused_s2.hip WAVES = 8).  Forced here on small ragged images, batches, UNORM storage and row bands: the image is the
    two dispatches', bit for bit."""
    iw, ih = shape
    ow, oh = 2 * iw, 2 * ih
    n = 2
    src = dev(np.stack([frames.synthetic_frame(iw, ih, k=60 + f, dtype=np.float16) for f in range(n)]))
    with fsr._lib.test_hooks() as lib:  # libfsr1_hip_test.so (include/fsr1_hip_test.h)
        lib.fsr1_debug_fused_tall_tiles(1)
        for flags in (0, fsr.FLAG_MATH_EXACT | fsr.FLAG_RCAS_DENOISE, fsr.FLAG_RCAS_PASSTHROUGH_ALPHA | fsr.FLAG_HDR_SQUARE):
            mid = torch.zeros(n, oh, ow, 4, dtype=torch.float16, device="cuda")
            two = torch.zeros_like(mid)
            fsr.easu(src, mid, flags=flags & fsr.FLAG_MATH_EXACT)
            fsr.rcas(mid, two, sharpness=0.3, flags=flags)
            big_out = torch.full((n, oh + 1, ow + 5, 4), 7, dtype=torch.float16, device="cuda")
            dst = big_out[:, :oh, :ow]
            fsr.easu_rcas_fused(src, dst, sharpness=0.3, flags=flags)
            torch.cuda.synchronize()
            assert bool((big_out[:, oh:] == 7).all()) and bool((big_out[:, :, ow:] == 7).all()), "wrote outside the output view"
            assert torch.equal(dst.view(torch.int16), two.view(torch.int16)), "tall tiles != two dispatches (flags %d)" % flags
            band = torch.full_like(two[0], -1.0)
            cuts = [0, (oh // 3) & ~1, (2 * oh // 3) & ~1, oh]
            for y0, y1 in zip(cuts, cuts[1:]):
                if y1 > y0:
                    fsr.upscale_band(src[0], band[y0:y1], (ow, oh), (y0, y1), sharpness=0.3, flags=flags, fused=True)
            assert torch.equal(band.view(torch.int16), two[0].view(torch.int16)), "bands with tall tiles != two dispatches (flags %d)" % flags
        # RGBA8 storage
        src8 = (src.float().clamp(0, 1) * 255 + 0.5).to(torch.uint8)
        mid8 = torch.zeros(n, oh, ow, 4, dtype=torch.uint8, device="cuda")
        two8 = torch.zeros_like(mid8)
        one8 = torch.zeros_like(mid8)
        fsr.easu(src8, mid8)
        fsr.rcas(mid8, two8, sharpness=0.3)
        fsr.easu_rcas_fused(src8, one8, sharpness=0.3)
        assert torch.equal(one8, two8)


def test_selftest_binary32_rcp_ieee_over_all_operands(fsr):
    """fsr1_selftest() also runs the EXACT variants' 6-instruction reciprocal rcp_ieee (include/fsr1_device_base.hpp) against the IEEE
    division 1.0f / x for ALL 2^32 binary32 operands on the device (selftest_rcp_kernel, csrc/fsr1_api.hip) — the measurement behind
    "EXACT is bit-identical to the reference's ARcpF1" — next to the exhaustive binary16 check of the H kernels' reciprocal
    (tests/test_gpu_parity_h.py): 0 differing operands."""
    import time
    t0 = time.perf_counter()
    assert fsr.selftest() == 0
    assert time.perf_counter() - t0 < 30.0  # 2^32 operands take about a second on an MI355X


@pytest.mark.parametrize("shape", [(97, 160), (31, 75), (64, 40), (70, 9), (1, 1), (200, 31), (33, 17)], ids=lambda s: "%dx%d" % s)
def test_easu_exact_2x_tall_tiles(fsr, shape):
    """Exact-2x EASU launches that are large, or that overlap other frames' launches, run on 64 x 32 tiles (easu_kernel<..., TH = 32>):
    forced here on small ragged images and batches — bit-identical to the 64 x 16 tiles and to the generic kernel, for the default
    and the EXACT arithmetic, RGBA16F / RGBA32F / RGBA8 storage, the HDR square, and with the overlap hint given by hand."""
    iw, ih = shape
    ow, oh = 2 * iw, 2 * ih
    n = 2
    src16 = dev(np.stack([frames.synthetic_frame(iw, ih, k=80 + f, dtype=np.float16) for f in range(n)]))
    for src in (src16, src16.float(), (src16.float().clamp(0, 1) * 255 + 0.5).to(torch.uint8)):
        for flags in (0, fsr.FLAG_MATH_EXACT, fsr.FLAG_HDR_SQUARE | fsr.FLAG_OUTPUT_STREAMING):
            with fsr._lib.test_hooks() as lib:  # libfsr1_hip_test.so (include/fsr1_hip_test.h)
                lib.fsr1_debug_easu_tall_tiles(0)
                want = torch.zeros(n, oh, ow, 4, dtype=src.dtype, device="cuda")
                fsr.easu(src, want, flags=flags)
                gen = torch.zeros_like(want)
                fsr.easu(src, gen, flags=flags | fsr.FLAG_NO_FAST_PATHS)
                lib.fsr1_debug_easu_tall_tiles(1)
                big = torch.full((n, oh + 1, ow + 5, 4), 7, dtype=src.dtype, device="cuda")
                got = big[:, :oh, :ow]
                fsr.easu(src, got, flags=flags)
                torch.cuda.synchronize()
            assert bool((big[:, oh:] == 7).all()) and bool((big[:, :, ow:] == 7).all()), "wrote outside the output view"
            assert torch.equal(got, want) and torch.equal(gen, want), (str(src.dtype), flags)
            hint = torch.zeros_like(want)
            fsr.easu(src, hint, flags=flags | fsr.FLAG_FRAMES_OVERLAP)  # the rule's own choice under the overlap hint: tall
            assert torch.equal(hint, want), (str(src.dtype), flags, "overlap hint")


@pytest.mark.parametrize("shape", [(97, 160, 146, 240), (120, 68, 204, 116), (200, 120, 260, 156), (64, 40, 69, 43), (33, 17, 57, 29), (300, 170, 450, 255)],
                         ids=lambda s: "%dx%d_to_%dx%d" % s)
def test_easu_generic_tall_tiles(fsr, shape):
    """Ratios without an exact-2x form run the generic kernel; large launches take its 512-thread workgroups on 64 x 32 tiles
    (easu_kernel<..., TH = 32, WAVES = 8>, round 5).  Forced here on small ragged images and batches at 1.5x / 1.7x / 1.3x / 1.08x — bit-
    identical to the 64 x 16 tiles of 256 threads, in RGBA16F / RGBA32F / RGBA8 storage and with the HDR square; and the rule's own choice
    on a frame large enough to take the tall tiles equals the short tiles as well."""
    iw, ih, ow, oh = shape
    n = 2
    src16 = dev(np.stack([frames.synthetic_frame(iw, ih, k=90 + f, dtype=np.float16) for f in range(n)]))
    for src in (src16, src16.float(), (src16.float().clamp(0, 1) * 255 + 0.5).to(torch.uint8)):
        for flags in (0, fsr.FLAG_HDR_SQUARE | fsr.FLAG_OUTPUT_STREAMING):
            with fsr._lib.test_hooks() as lib:  # libfsr1_hip_test.so (include/fsr1_hip_test.h)
                lib.fsr1_debug_easu_tall_tiles(0)
                want = torch.zeros(n, oh, ow, 4, dtype=src.dtype, device="cuda")
                fsr.easu(src, want, con=fsr.FsrEasuCon(iw, ih, iw, ih, ow, oh), flags=flags)
                lib.fsr1_debug_easu_tall_tiles(1)
                big = torch.full((n, oh + 1, ow + 5, 4), 7, dtype=src.dtype, device="cuda")
                got = big[:, :oh, :ow]
                fsr.easu(src, got, con=fsr.FsrEasuCon(iw, ih, iw, ih, ow, oh), flags=flags)
                torch.cuda.synchronize()
            assert bool((big[:, oh:] == 7).all()) and bool((big[:, :, ow:] == 7).all()), "wrote outside the output view"
            assert torch.equal(got, want), (str(src.dtype), flags)


def test_easu_generic_tall_tiles_rule_on_a_large_frame(fsr):
    """2560x1440 -> 3840x2160: the host's rule takes the 512-thread tiles (2040 workgroups); the same frame with the rule overridden to
    the 256-thread tiles is the same image, and so is the EXACT-free generic path of the two-dispatch pipeline (RCAS on top)."""
    iw, ih, ow, oh = 2560, 1440, 3840, 2160
    src = dev(frames.synthetic_frame(iw, ih, k=5, dtype=np.float16))
    con = fsr.FsrEasuCon(iw, ih, iw, ih, ow, oh)
    rule = torch.zeros(oh, ow, 4, dtype=torch.float16, device="cuda")
    fsr.easu(src, rule, con=con)
    with fsr._lib.test_hooks() as lib:
        lib.fsr1_debug_easu_tall_tiles(0)
        short = torch.zeros_like(rule)
        fsr.easu(src, short, con=con)
        lib.fsr1_debug_easu_tall_tiles(1)
        tall = torch.zeros_like(rule)
        fsr.easu(src, tall, con=con)
        torch.cuda.synchronize()
    assert torch.equal(rule.view(torch.int16), short.view(torch.int16)) and torch.equal(tall.view(torch.int16), short.view(torch.int16))
