"""Colour stages on the GPU: FsrSrtmF / FsrLfgaF / FsrSrtmInvF / FsrTepdC8F|C10F (ffx_fsr1.h:986-1199)
as a stand-alone pass (fsr1_color_dispatch) and as fused prologue / epilogue of EASU, RCAS and the fused kernel
(fsr1_*_dispatch_ex), through the C ABI, against the CPU oracle and the committed golden vectors.

Bars:
  EXACT (FSR1_FLAG_MATH_EXACT)  every stored value equals the oracle's — binary32 bits for RGBA32F, binary16 bits for
                                RGBA16F, codes for the UNORM formats;
  default                       the two ARcpF1 reciprocals are v_rcp_f32 (1 ulp): within 1 binary16 ULP / 1 code,
                                >= 99.5 % equal.  FsrLfgaF and FsrTepd*F contain no reciprocal and are bit-exact in
                                both modes (checked as such where they run alone).
"""
import importlib

import numpy as np
import pytest

from conftest import load_golden
from test_gpu_parity import assert_exact16, assert_exact32, assert_f_class, dev, host
from test_gpu_unorm import check_codes, decode, encode, pack10

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
frames = importlib.import_module("fidelityfx-fsr_amd.frames")
PARAMS = dict(amount=0.75, bias=0.0, frame=3, noise_offset=(5, -3))  # tests/golden/gen_golden.py COLOR_PARAMS


def stages_of(fsr, st, noise_t, p=PARAMS):
    return fsr.ColorStages(st, grain_amount=p["amount"], grain_bias=p["bias"], frame=p["frame"], noise=noise_t,
                           noise_offset=p["noise_offset"])


def test_color_pass_matches_golden(fsr):
    """Stand-alone pass, RGBA32F in/out, EXACT: bit-identical to the CPU-evaluated reference on the committed vectors."""
    g = load_golden("color_stages")
    noise = dev(g["noise"].astype(np.float32))
    n = 0
    for k in sorted(g):
        if not k.startswith("out_"):
            continue
        _, st, key = k.split("_")
        src = dev(g[key].astype(np.float32))
        dst = torch.zeros_like(src)
        fsr.color(src, dst, stages_of(fsr, int(st), noise), flags=fsr.FLAG_MATH_EXACT)
        assert_exact32(host(dst), g[k], "stages %s on %s" % (st, key))
        n += 1
    assert n == 12


@pytest.mark.parametrize("stages", [1, 2, 4, 8, 16, 8 | 32, 2 | 8, 2 | 4, 1 | 2 | 4, 2 | 16 | 32])
@pytest.mark.parametrize("exact", [True, False], ids=["exact", "f"])
def test_color_pass_rgba16f(fsr, port, stages, exact):
    """RGBA16F in/out on a ragged size (odd width: single-pixel tail; height not a multiple of the block), fp16 noise
    with several slices and negative offsets."""
    rng = np.random.default_rng(100 + stages)
    h, w = 77, 203
    img = rng.random((h, w, 4)).astype(np.float32)
    if stages & 1:
        # FsrSrtmInvF divides by 1 - max3(c): a peak of P amplifies the 1-ulp v_rcp_f32 of FsrSrtmF by ~P, so the
        # default-arithmetic 1-ULP bar is meaningful for the round trip only while P stays well below 2^12
        img[..., :3] = img[..., :3] ** 4 * (20000.0 if exact or not stages & 4 else 60.0)
    img = img.astype(np.float16)
    noise = (rng.random((3, 16, 24, 4)).astype(np.float32) - np.array([0.5, 0.5, 0.5, 0.0], np.float32)).astype(np.float16)
    p = dict(amount=0.6, bias=0.05, frame=7, noise_offset=(-9, 1000))
    want = port.color_f(img.astype(np.float32), stages, noise=noise.astype(np.float32), **p)
    dst = torch.zeros(h, w, 4, dtype=torch.float16, device="cuda")
    fsr.color(dev(img), dst, stages_of(fsr, stages, dev(noise), p), flags=fsr.FLAG_MATH_EXACT if exact else 0)
    got = host(dst)
    if exact or not (stages & 5):  # no reciprocal in LFGA / TEPD: bit-exact in both modes
        assert_exact16(got, want, "stages %d" % stages)
    else:
        assert_f_class(got, want, "stages %d" % stages)


@pytest.mark.parametrize("fmt", ["rgba8", "rgb10a2"])
def test_color_pass_tepd_to_unorm(fsr, port, fmt):
    """The TEPD targets: RGBA16F linear -> dithered gamma-2.0 codes in RGBA8 / R10G10B10A2 (FsrTepdC8F / C10F).
    The stage's output is on the code grid, so the store conversion reproduces it exactly; bit-exact in both modes."""
    h, w = 120, 200
    img = frames.synthetic_frame(w, h, k=3, dtype=np.float32).astype(np.float16)
    img[0, :4, :3] = np.array([0.0, 1.0, 0.5, 1e-6], np.float16)[:, None]
    st = 8 if fmt == "rgba8" else 16
    want = port.color_f(img.astype(np.float32), st, frame=11)
    for flags in (0, fsr.FLAG_MATH_EXACT):
        if fmt == "rgba8":
            dst = torch.zeros(h, w, 4, dtype=torch.uint8, device="cuda")
            fsr.color(dev(img), dst, fsr.ColorStages(st, frame=11), flags=flags)
            check_codes(host(dst), encode(want, 255), True, "tepd c8")
            # on the grid: decode(code) is within an ulp of the stage's output
            np.testing.assert_allclose(decode(host(dst)[..., :3], 255), want[..., :3], rtol=0, atol=1e-6)
        else:
            dst = torch.zeros(h, w, dtype=torch.int32, device="cuda")
            fsr.color(dev(img), dst, fsr.ColorStages(st, frame=11), flags=flags)
            assert np.array_equal(host(dst).view(np.uint32), pack10(want))


def test_color_pass_in_place_and_batch(fsr, port):
    """frames > 1 with pitches, in place (in == out)."""
    rng = np.random.default_rng(5)
    n, h, w = 3, 33, 50
    buf = torch.zeros(n, h + 2, w + 6, 4, dtype=torch.float32, device="cuda")
    img = rng.random((n, h, w, 4)).astype(np.float32)
    view = buf[:, :h, :w]
    view.copy_(dev(img))
    noise = (rng.random((1, 8, 8, 4)) - 0.5).astype(np.float32)
    fsr.color(view, view, fsr.ColorStages(2 | 4, grain_amount=0.5, noise=dev(noise)), flags=fsr.FLAG_MATH_EXACT)
    got = host(view)
    for f in range(n):
        assert_exact32(got[f], port.color_f(img[f], 2 | 4, amount=0.5, noise=noise), "frame %d" % f)
    assert float(buf[:, h:].abs().sum()) == 0.0 and float(buf[:, :, w:].abs().sum()) == 0.0  # padding untouched


def pipeline_oracle(port, img_f32, ow, oh, con, rcon, st, p, noise, mid_round, rcas_flags=0):
    """CPU restatement of the fused pipeline: prologue on the input texels, EASU, intermediary rounded to the storage
    format, RCAS, epilogue."""
    pre = port.color_f(img_f32, st & 1) if st & 1 else img_f32
    mid = mid_round(port.easu_f(pre, ow, oh, con))
    out = port.rcas_f(mid, rcon, rcas_flags)
    post = st & ~1
    return port.color_f(out, post, noise=noise, **p) if post else out


@pytest.mark.parametrize("stages", [1, 2, 1 | 2 | 4, 2 | 8, 1 | 4, 2 | 16 | 32])
@pytest.mark.parametrize("exact", [True, False], ids=["exact", "f"])
def test_fused_with_colour_stages(fsr, port, stages, exact):
    """fsr1_easu_rcas_fused_dispatch_ex: SRTM at the EASU loads, LFGA / SRTM_INV / TEPD after RCAS, RGBA16F.
    EXACT reproduces the oracle pipeline bit for bit; it also equals the two *_ex dispatches."""
    iw, ih, ow, oh = 150, 85, 300, 170
    rng = np.random.default_rng(stages)
    img = frames.synthetic_frame(iw, ih, k=4, dtype=np.float32)
    if stages & 1:
        img[..., :3] = img[..., :3] ** 3 * 500.0
    img = img.astype(np.float16)
    noise = (rng.random((2, 32, 32, 4)).astype(np.float32) - np.array([0.5, 0.5, 0.5, 0.0], np.float32)).astype(np.float16)
    p = dict(amount=0.5, bias=0.0, frame=2, noise_offset=(3, 4))
    con = port.FsrEasuCon(iw, ih, iw, ih, ow, oh)
    rcon = port.FsrRcasCon(0.25)
    flags = fsr.FLAG_MATH_EXACT if exact else 0
    cs = stages_of(fsr, stages, dev(noise), p)
    src = dev(img)
    dst = torch.zeros(oh, ow, 4, dtype=torch.float16, device="cuda")
    fsr.easu_rcas_fused(src, dst, easu_con=con, rcas_con=rcon, flags=flags, stages=cs)
    got = host(dst)
    # two-pass with the same stages: prologue on EASU, epilogue on RCAS
    mid = torch.zeros(oh, ow, 4, dtype=torch.float16, device="cuda")
    two = torch.zeros(oh, ow, 4, dtype=torch.float16, device="cuda")
    pre = fsr.ColorStages(stages & 1) if stages & 1 else None
    post = stages_of(fsr, stages & ~1, dev(noise), p) if stages & ~1 else None
    fsr.easu(src, mid, con=con, flags=flags, stages=pre)
    fsr.rcas(mid, two, con=rcon, flags=flags, stages=post)
    assert np.array_equal(host(two).view(np.uint16), got.view(np.uint16)), "fused_ex differs from easu_ex + rcas_ex"
    r16 = lambda a: a.astype(np.float16).astype(np.float32)
    if exact:
        want = pipeline_oracle(port, img.astype(np.float32), ow, oh, con, rcon, stages, p, noise.astype(np.float32), r16)
        assert_exact16(got, want, "fused stages %d" % stages)
    else:
        # stage-wise: EASU (with prologue) within 1 ULP of the oracle; the rest checked on the GPU's own intermediary
        pre_img = port.color_f(img.astype(np.float32), 1) if stages & 1 else img.astype(np.float32)
        assert_f_class(host(mid), port.easu_f(pre_img, ow, oh, con), "easu_ex")
        if not stages & (8 | 16):  # continuous epilogue: 1 ULP
            out = port.rcas_f(host(mid).astype(np.float32), rcon)
            want = port.color_f(out, stages & ~1, noise=noise.astype(np.float32), **p) if stages & ~1 else out
            assert_f_class(got, want, "rcas_ex")


@pytest.mark.parametrize("fmt", ["rgba8", "rgb10a2"])
def test_fused_tepd_to_unorm(fsr, port, fmt):
    """The SDR chain in one launch: RGBA16F linear in -> EASU -> RCAS -> film grain -> TEPD -> 8 / 10-bit codes out."""
    iw, ih, ow, oh = 128, 72, 256, 144
    img = frames.synthetic_frame(iw, ih, k=8, dtype=np.float16)
    rng = np.random.default_rng(1)
    noise = (rng.random((1, 16, 16, 4)).astype(np.float32) - np.array([0.5, 0.5, 0.5, 0.0], np.float32)).astype(np.float16)
    p = dict(amount=0.3, bias=0.0, frame=9, noise_offset=(0, 0))
    st = 2 | (8 if fmt == "rgba8" else 16)
    con = port.FsrEasuCon(iw, ih, iw, ih, ow, oh)
    rcon = port.FsrRcasCon(0.25)
    r16 = lambda a: a.astype(np.float16).astype(np.float32)
    want = pipeline_oracle(port, img.astype(np.float32), ow, oh, con, rcon, st, p, noise.astype(np.float32), r16)
    cs = stages_of(fsr, st, dev(noise), p)
    if fmt == "rgba8":
        dst = torch.zeros(oh, ow, 4, dtype=torch.uint8, device="cuda")
        fsr.easu_rcas_fused(dev(img), dst, easu_con=con, rcas_con=rcon, flags=fsr.FLAG_MATH_EXACT, stages=cs)
        check_codes(host(dst), encode(want, 255), True, "fused tepd c8")
        dflt = torch.zeros_like(dst)
        fsr.easu_rcas_fused(dev(img), dflt, easu_con=con, rcas_con=rcon, flags=0, stages=cs)
        # default arithmetic: a 1-ULP difference upstream can move a pixel across a dither threshold: one code, rarely
        d = np.abs(host(dflt).astype(np.int32) - host(dst).astype(np.int32))
        assert d.max() <= 1 and (d == 0).mean() > 0.99
    else:
        dst = torch.zeros(oh, ow, dtype=torch.int32, device="cuda")
        fsr.easu_rcas_fused(dev(img), dst, easu_con=con, rcas_con=rcon, flags=fsr.FLAG_MATH_EXACT, stages=cs)
        assert np.array_equal(host(dst).view(np.uint32), pack10(want))


def test_easu_only_with_epilogue(fsr, port):
    """EASU as the last pass (bUseRcas off): prologue + epilogue on fsr1_easu_dispatch_ex, RGBA32F, EXACT."""
    iw, ih, ow, oh = 61, 35, 122, 70
    img = frames.synthetic_frame(iw, ih, k=2, dtype=np.float32)
    img[..., :3] = img[..., :3] ** 2 * 50.0
    con = port.FsrEasuCon(iw, ih, iw, ih, ow, oh)
    dst = torch.zeros(oh, ow, 4, dtype=torch.float32, device="cuda")
    fsr.easu(dev(img), dst, con=con, flags=fsr.FLAG_MATH_EXACT, stages=fsr.ColorStages(1 | 4))
    want = port.color_f(port.easu_f(port.color_f(img, 1), ow, oh, con), 4)
    assert_exact32(host(dst), want, "easu_ex srtm/inv")


def test_colour_stage_argument_validation(fsr):
    src = torch.zeros(8, 8, 4, dtype=torch.float16, device="cuda")
    dst = torch.zeros(8, 8, 4, dtype=torch.float16, device="cuda")
    with pytest.raises(fsr.Fsr1Error, match="noise"):
        fsr.color(src, dst, fsr.ColorStages(2))                     # LFGA without a noise image
    with pytest.raises(fsr.Fsr1Error, match="exclusive"):
        fsr.color(src, dst, fsr.ColorStages(8 | 16))
    with pytest.raises(fsr.Fsr1Error, match="unknown colour stage"):
        fsr.color(src, dst, fsr.ColorStages(1 << 9))
    f32 = torch.zeros(8, 8, 4, dtype=torch.float32, device="cuda")
    with pytest.raises(fsr.Fsr1Error, match="format pair"):
        fsr.rcas(src, f32, stages=fsr.ColorStages(4))               # RGBA16F -> RGBA32F is not a built pair
    with pytest.raises(fsr.Fsr1Error, match="format"):
        fsr.rcas(src, f32)                                            # and never without stages


@pytest.mark.parametrize("fused", [False, True], ids=["two-pass", "fused"])
def test_upscale_with_colour_stages(fsr, port, fused):
    """FSR_Filter::Upscale with stages (fsr1_upscale_ex): prologue on EASU, epilogue on the pass that writes the output;
    two-pass and fused give the same image, EXACT reproduces the oracle pipeline; with bUseRcas off EASU takes both."""
    iw, ih, ow, oh = 96, 54, 192, 108
    img = (frames.synthetic_frame(iw, ih, k=1, dtype=np.float32) ** 2 * 30.0).astype(np.float16)
    img[..., 3] = 1
    noise = ((np.random.default_rng(4).random((1, 8, 8, 4)) - 0.5) * np.array([1, 1, 1, 0]) + np.array([0, 0, 0, 0.5])).astype(np.float16)
    p = dict(amount=0.4, bias=0.0, frame=0, noise_offset=(0, 0))
    st = 1 | 2 | 4
    cs = stages_of(fsr, st, dev(noise), p)
    filt = fsr.FSR_Filter()
    filt.OnCreate(slowFallback=True, exact=True, fused=fused)
    dst = torch.zeros(oh, ow, 4, dtype=torch.float16, device="cuda")
    filt.OnCreateWindowSizeDependentResources(dev(img), dst, ow, oh)
    filt.Upscale(ow, oh, fsr.State(iw, ih, bUseRcas=True, rcasAttenuation=0.25), stages=cs)
    con = port.FsrEasuCon(iw, ih, iw, ih, ow, oh)
    r16 = lambda a: a.astype(np.float16).astype(np.float32)
    want = pipeline_oracle(port, img.astype(np.float32), ow, oh, con, port.FsrRcasCon(0.25), st, p, noise.astype(np.float32), r16)
    assert_exact16(host(dst), want, "Upscale + stages")
    # EASU only
    filt.Upscale(ow, oh, fsr.State(iw, ih, bUseRcas=False), stages=cs)
    pre = port.color_f(img.astype(np.float32), 1)
    want = port.color_f(port.easu_f(pre, ow, oh, con), st & ~1, noise=noise.astype(np.float32), **p)
    assert_exact16(host(dst), want, "Upscale (EASU only) + stages")
    filt.OnDestroy()


@pytest.mark.parametrize("stages", [1, 2, 4, 8, 16, 8 | 32, 2 | 8, 2 | 4, 1 | 2 | 4, 2 | 16 | 32])
def test_color_pass_packed_fp16(fsr, port, stages):
    """FSR1_FLAG_MATH_PACKED_FP16: FsrSrtmH / FsrLfgaH / FsrSrtmInvH / FsrTepdC8H | C10H (Hx2 forms, two pixels per AH2) —
    parity class H: every binary16 output bit equals the CPU-evaluated half-precision path; ragged size, fp16 noise."""
    rng = np.random.default_rng(200 + stages)
    h, w = 45, 131
    img = rng.random((h, w, 4)).astype(np.float32)
    if stages & 1:
        img[..., :3] = img[..., :3] ** 4 * 20000.0
    img[0, :5, :3] = np.array([0.0, 1.0, 0.5, 6e-8, 0.25], np.float32)[:, None]
    img = img.astype(np.float16)
    noise = (rng.random((3, 16, 24, 4)).astype(np.float32) - np.array([0.5, 0.5, 0.5, 0.0], np.float32)).astype(np.float16)
    p = dict(amount=0.6, bias=0.05, frame=7, noise_offset=(-9, 1000))
    want = port.color_h(img.astype(np.float32), stages, noise=noise.astype(np.float32), **p)
    dst = torch.zeros(h, w, 4, dtype=torch.float16, device="cuda")
    fsr.color(dev(img), dst, stages_of(fsr, stages, dev(noise), p), flags=fsr.FLAG_MATH_PACKED_FP16)
    assert_exact16(host(dst), want, "H stages %d" % stages)


def test_color_pass_packed_fp16_golden(fsr):
    g = load_golden("color_stages")
    noise = dev(g["noise"])
    n = 0
    for k in sorted(g):
        if not k.startswith("outh_"):
            continue
        _, st, key = k.split("_")
        src = dev(g[key])
        dst = torch.zeros_like(src)
        fsr.color(src, dst, stages_of(fsr, int(st), noise), flags=fsr.FLAG_MATH_PACKED_FP16)
        assert_exact16(host(dst), g[k].astype(np.float32), "H golden stages %s on %s" % (st, key))
        n += 1
    assert n == 12
