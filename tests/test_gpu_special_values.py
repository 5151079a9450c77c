"""Hostile but finite inputs on the GPU (frames.adversarial_frame: HDR highlights up to 65504, zeros, negative texels,
binary16 subnormals, single-channel spikes).  The oracle is pinned on the same inputs by tests/test_special_values.py.

  EXACT : bit-identical to the CPU-evaluated FsrEasuF / FsrRcasF, Inf and NaN included (same places; NaN payloads are free),
          for RGBA32F and RGBA16F storage, two dispatches and the fused launch, exact-2x and generic kernels;
  H     : FsrEasuH / FsrRcasH bit-identical to the H oracle — binary16 overflow to Inf and the NaNs that follow included;
  F     : default arithmetic — non-finite values at the same places as the oracle, >= 99.9 % of the finite values within
          1 binary16 ULP and >= 99.5 % bit-equal.  The <= 1 ULP bound of class F is a statement about image-like content
          (asserted on whole frames elsewhere): where a 12-tap window mixes magnitudes 1e4 .. 1e10 apart, the negative lobes
          cancel terms far larger than the result and the re-associated weights (1e-7 relative) show up as a few ULP on about
          one value in 1e4 (tools/experiments_r02/outlier_probe.py prints them); FSR1_FLAG_MATH_EXACT is the bit-exact path.
"""
import importlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
frames = importlib.import_module("fidelityfx-fsr_amd.frames")

SHAPES = [(96, 54, 192, 108), (80, 45, 120, 68), (61, 35, 79, 46)]  # exact 2x (quad kernel), 1.5x, odd ratio


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def host(t):
    torch.cuda.synchronize()
    return t.detach().cpu().numpy()


def same(got, want):
    """bit-identical as arrays of got's dtype, any NaN equal to any NaN"""
    got = np.asarray(got)
    with np.errstate(over="ignore"):
        want = np.asarray(want, np.float32).astype(got.dtype)
    u = {2: np.uint16, 4: np.uint32}[got.dtype.itemsize]
    bad = (got.view(u) != want.view(u)) & ~(np.isnan(got) & np.isnan(want))
    return int(bad.sum()), (np.argwhere(bad)[:3].tolist(), got[bad][:3], want[bad][:3])


def assert_same(got, want, what):
    n, first = same(got, want)
    assert n == 0, "%s: %d values differ, first %s" % (what, n, first)


def assert_f_class_with_specials(got16, want_f32, what):
    import cpu_oracle
    got = np.asarray(got16, np.float32)
    with np.errstate(over="ignore"):  # binary16 overflow to Inf is part of the comparison
        want16 = np.asarray(want_f32, np.float32).astype(np.float16).astype(np.float32)
    special = ~np.isfinite(want16)
    assert np.array_equal(np.isnan(got), np.isnan(want16)), what + ": NaNs at different places"
    assert np.array_equal(got[special & ~np.isnan(want16)], want16[special & ~np.isnan(want16)]), what + ": infinities differ"
    d = cpu_oracle.half_ulp_diff(got[~special], want16[~special])
    print("%s: ulp histogram %s of %d" % (what, np.bincount(np.minimum(d, 8)).tolist(), d.size))
    assert float((d <= 1).mean()) >= 0.999, "%s: only %.5f of the finite values within 1 binary16 ULP" % (what, float((d <= 1).mean()))
    assert d.max() <= 64, "%s: max %d binary16 ULP" % (what, d.max())
    return float((d == 0).mean())


@pytest.mark.parametrize("shape", SHAPES, ids=lambda s: "%dx%d_to_%dx%d" % s)
def test_exact_arithmetic_on_adversarial_values(fsr, port, shape):
    iw, ih, ow, oh = shape
    con = port.FsrEasuCon(iw, ih, iw, ih, ow, oh)
    rcon = port.FsrRcasCon(0.25)
    for k in (0, 1):
        img = frames.adversarial_frame(iw, ih, k=k, dtype=np.float32)
        want_e = port.easu_f(img, ow, oh, con)
        for dt in (torch.float32, torch.float16):
            src = dev(img).to(dt)
            mid = torch.zeros(oh, ow, 4, dtype=dt, device="cuda")
            fsr.easu(src, mid, con=con, flags=fsr.FLAG_MATH_EXACT)
            assert_same(host(mid), want_e, "easu EXACT %s k=%d" % (dt, k))
            if shape[2] == 2 * shape[0]:
                gen = torch.zeros_like(mid)
                fsr.easu(src, gen, con=con, flags=fsr.FLAG_MATH_EXACT | fsr.FLAG_NO_FAST_PATHS)
                assert torch.equal(gen.view(torch.int32 if dt == torch.float32 else torch.int16), mid.view(torch.int32 if dt == torch.float32 else torch.int16))
            # RCAS on what EASU stored (binary16 storage rounds the intermediary, as the two-pass pipeline does)
            mid_host = host(mid).astype(np.float32)
            for fl, bits in ((0, 0), (1, fsr.FLAG_RCAS_DENOISE), (3, fsr.FLAG_RCAS_DENOISE | fsr.FLAG_RCAS_PASSTHROUGH_ALPHA)):
                out = torch.zeros_like(mid)
                fsr.rcas(mid, out, con=rcon, flags=bits | fsr.FLAG_MATH_EXACT)
                assert_same(host(out), port.rcas_f(mid_host, rcon, fl), "rcas EXACT %s k=%d flags %d" % (dt, k, fl))
            out = torch.zeros_like(mid)
            fsr.rcas(mid, out, con=rcon, flags=fsr.FLAG_MATH_EXACT)
            fused = torch.zeros_like(mid)
            fsr.easu_rcas_fused(src, fused, easu_con=con, rcas_con=rcon, flags=fsr.FLAG_MATH_EXACT)
            n, first = same(host(fused), host(out).astype(np.float32))
            assert n == 0, "fused EXACT %s k=%d differs from the two dispatches: %s" % (dt, k, first)
        # RCAS fed hostile values directly (EASU's dering clamp keeps its output inside the local input range)
        hostile = frames.adversarial_frame(ow, oh, k=k + 7, dtype=np.float32)
        for dt in (torch.float32, torch.float16):
            t = dev(hostile).to(dt)
            for fl, bits in ((0, 0), (1, fsr.FLAG_RCAS_DENOISE), (4, fsr.FLAG_HDR_SQUARE)):
                out = torch.zeros_like(t)
                fsr.rcas(t, out, con=rcon, flags=bits | fsr.FLAG_MATH_EXACT)
                assert_same(host(out), port.rcas_f(hostile, rcon, fl), "rcas EXACT on hostile input %s k=%d flags %d" % (dt, k, fl))


@pytest.mark.parametrize("shape", SHAPES, ids=lambda s: "%dx%d_to_%dx%d" % s)
def test_packed_fp16_on_adversarial_values(fsr, port, shape):
    iw, ih, ow, oh = shape
    con = port.FsrEasuCon(iw, ih, iw, ih, ow, oh)
    rcon = port.FsrRcasCon(0.25)
    for k in (0, 1):
        img = frames.adversarial_frame(iw, ih, k=k, dtype=np.float32)
        src = dev(img).to(torch.float16)
        mid = torch.zeros(oh, ow, 4, dtype=torch.float16, device="cuda")
        fsr.easu(src, mid, con=con, flags=fsr.FLAG_MATH_PACKED_FP16)
        assert_same(host(mid), port.easu_h(img, ow, oh, con), "easu H k=%d" % k)
        hostile = frames.adversarial_frame(ow, oh, k=k + 7, dtype=np.float32)
        t = dev(hostile).to(torch.float16)
        for fl, bits in ((0, 0), (1, fsr.FLAG_RCAS_DENOISE), (2, fsr.FLAG_RCAS_PASSTHROUGH_ALPHA)):
            out = torch.zeros_like(t)
            fsr.rcas(t, out, con=rcon, flags=bits | fsr.FLAG_MATH_PACKED_FP16)
            assert_same(host(out), port.rcas_h(hostile, rcon, fl), "rcas H on hostile input k=%d flags %d" % (k, fl))
        two = torch.zeros_like(mid)
        fsr.rcas(mid, two, con=rcon, flags=fsr.FLAG_MATH_PACKED_FP16)
        fused = torch.zeros_like(mid)
        fsr.easu_rcas_fused(src, fused, easu_con=con, rcas_con=rcon, flags=fsr.FLAG_MATH_PACKED_FP16)
        n, first = same(host(fused), host(two).astype(np.float32))
        assert n == 0, "fused H k=%d differs from the two H dispatches: %s" % (k, first)


@pytest.mark.parametrize("shape", SHAPES, ids=lambda s: "%dx%d_to_%dx%d" % s)
def test_default_arithmetic_on_adversarial_values(fsr, port, shape, record_property):
    iw, ih, ow, oh = shape
    con = port.FsrEasuCon(iw, ih, iw, ih, ow, oh)
    rcon = port.FsrRcasCon(0.25)
    img = frames.adversarial_frame(iw, ih, k=2, dtype=np.float32)
    src = dev(img).to(torch.float16)
    mid = torch.zeros(oh, ow, 4, dtype=torch.float16, device="cuda")
    fsr.easu(src, mid, con=con)
    fe = assert_f_class_with_specials(host(mid), port.easu_f(img, ow, oh, con), "easu F")
    hostile = frames.adversarial_frame(ow, oh, k=9, dtype=np.float32)
    t = dev(hostile).to(torch.float16)
    out = torch.zeros_like(t)
    fsr.rcas(t, out, con=rcon)
    fr = assert_f_class_with_specials(host(out), port.rcas_f(hostile, rcon, 0), "rcas F")
    record_property("easu_bit_equal_fraction", fe)
    record_property("rcas_bit_equal_fraction", fr)
    print("adversarial %s: bit-equal fraction easu %.4f rcas %.4f" % (shape, fe, fr))
    assert fe >= 0.995 and fr >= 0.995


@pytest.mark.parametrize("stages", [1, 4, 1 | 4, 2, 1 | 2 | 4, 8, 16, 2 | 8])
def test_colour_stages_on_adversarial_values(fsr, port, stages):
    """The colour stages (ffx_fsr1.h:986-1199) on hostile values: EXACT bit-identical in RGBA32F and RGBA16F, the packed-fp16
    entry points bit-identical to the H oracle (negative and > 1 inputs drive FsrSrtmInvF through its 1/32768 floor, the tone
    mapper through negative peaks)."""
    rng = np.random.default_rng(77)
    noise = (rng.random((2, 8, 12, 4)).astype(np.float32) - np.array([0.5, 0.5, 0.5, 0.0], np.float32)).astype(np.float16)
    p = dict(amount=0.6, bias=0.05, frame=5, noise_offset=(-3, 17))
    st = fsr.ColorStages(stages, grain_amount=p["amount"], grain_bias=p["bias"], frame=p["frame"], noise=dev(noise), noise_offset=p["noise_offset"])
    img = frames.adversarial_frame(70, 41, k=0, dtype=np.float32)
    want = port.color_f(img, stages, noise=noise.astype(np.float32), **p)
    for dt in (torch.float32, torch.float16):
        src = dev(img).to(dt)
        dst = torch.zeros_like(src)
        fsr.color(src, dst, st, flags=fsr.FLAG_MATH_EXACT)
        assert_same(host(dst), want, "colour stages %d EXACT %s" % (stages, dt))
    src = dev(img).to(torch.float16)
    dst = torch.zeros_like(src)
    fsr.color(src, dst, st, flags=fsr.FLAG_MATH_PACKED_FP16)
    assert_same(host(dst), port.color_h(img, stages, noise=noise.astype(np.float32), **p), "colour stages %d H" % stages)
