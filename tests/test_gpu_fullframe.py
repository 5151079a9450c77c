"""Whole-frame parity on every BASELINE.json config, against the reference itself when it travelled.

The checker is `oracle/_ref/libfsr1_ref.so` — the reference headers compiled verbatim (oracle/build_ref.sh) — whenever
that library is present (it is built in the container and travels to the GPU box with the snapshot); only if it is
missing does the plain-C restatement (`port`, bit-exact against `_ref` per tests/test_oracle.py) stand in.  Every
row and column of every frame is compared, not bands:

    EXACT   FSR1_FLAG_MATH_EXACT            0 differing values (binary32 bits for RGBA32F, binary16 bits for RGBA16F)
    F       default arithmetic              <= 1 binary16 ULP, >= 99.5 % of the values bit-equal
    H       FSR1_FLAG_MATH_PACKED_FP16      bit-exact against the CPU-evaluated FsrEasuH / FsrRcasH
    UNORM   RGBA8 storage, EXACT            every code equal

RCAS is always judged on identical input (the GPU's own EASU output), SURVEY.md Appendix A.  Cites:
ffx_fsr1.h:315-437 (FsrEasuF), :505-593 (FsrEasuH), :684-769 (FsrRcasF), :782-866 (FsrRcasH).
"""
import importlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
frames = importlib.import_module("fidelityfx-fsr_amd.frames")

ULP_TOL = 1
MIN_EXACT_FRACTION = 0.995


@pytest.fixture(scope="module")
def checker():
    """oracle/_ref when present (kind "reference"), else the restatement (kind "port")."""
    import cpu_oracle
    return cpu_oracle.ref() if cpu_oracle.have_ref() else cpu_oracle.port()


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def host(t):
    torch.cuda.synchronize()
    return t.detach().cpu().numpy()


def h16(a):
    return np.ascontiguousarray(a).astype(np.float16).view(np.uint16)


def assert_exact16(gpu_f16, want_f32, what):
    g, o = h16(gpu_f16), h16(want_f32)
    nan = np.isnan(np.asarray(gpu_f16, np.float32)) & np.isnan(np.asarray(want_f32, np.float32))
    bad = (g != o) & ~nan
    assert not bad.any(), "%s: %d of %d binary16 values differ (first at %s)" % (what, bad.sum(), bad.size, np.argwhere(bad)[:3].tolist())


def assert_exact32(gpu_f32, want_f32, what):
    g = np.ascontiguousarray(gpu_f32, np.float32).view(np.uint32)
    o = np.ascontiguousarray(want_f32, np.float32).view(np.uint32)
    bad = (g != o) & ~(np.isnan(gpu_f32) & np.isnan(want_f32))
    assert not bad.any(), "%s: %d of %d binary32 values differ (first at %s)" % (what, bad.sum(), bad.size, np.argwhere(bad)[:3].tolist())


def assert_f_class(gpu, want_f32, what):
    import cpu_oracle
    g = np.asarray(gpu, np.float32)
    d = cpu_oracle.half_ulp_diff(g, want_f32)
    assert d.max() <= ULP_TOL, "%s: max %d binary16 ULP (tolerance %d) at %s" % (what, d.max(), ULP_TOL, np.argwhere(d > ULP_TOL)[:3].tolist())
    frac = float((d == 0).mean())
    assert frac >= MIN_EXACT_FRACTION, "%s: only %.4f of the values bit-equal" % (what, frac)
    assert not np.isnan(g).any(), what + ": NaN in the output"


def _mono(h16):
    """binary16 bit patterns (int16 view) -> integers monotone in the value (sign-magnitude unfolded), on the GPU"""
    i = h16.view(torch.int16).to(torch.int32)
    return torch.where(i < 0, -(i & 0x7FFF), i)


def gpu_assert_exact16(got16, want_f32, what):
    """assert_exact16 evaluated on the device: `want` is uploaded as binary32 and rounded RTNE there (same rounding as numpy)"""
    want16 = torch.from_numpy(np.ascontiguousarray(want_f32, np.float32)).cuda().to(torch.float16)
    bad = got16.view(torch.int16) != want16.view(torch.int16)
    n = int(bad.sum())
    assert n == 0, "%s: %d of %d binary16 values differ (first at %s)" % (what, n, bad.numel(), torch.nonzero(bad)[:3].tolist())


def gpu_assert_f_class(got16, want_f32, what):
    want16 = torch.from_numpy(np.ascontiguousarray(want_f32, np.float32)).cuda().to(torch.float16)
    d = (_mono(got16) - _mono(want16)).abs()
    mx = int(d.max())
    assert mx <= ULP_TOL, "%s: max %d binary16 ULP (tolerance %d)" % (what, mx, ULP_TOL)
    frac = float((d == 0).float().mean())
    assert frac >= MIN_EXACT_FRACTION, "%s: only %.4f of the values bit-equal" % (what, frac)
    assert not bool(torch.isnan(got16).any()), what + ": NaN in the output"


SHAPES = {
    "540p_to_1080p": (960, 540, 1920, 1080),     # configs[0] shape
    "1080p_to_4k": (1920, 1080, 3840, 2160),     # configs[1], [3]
    "1440p_to_4k": (2560, 1440, 3840, 2160),     # configs[2]
    "4k_to_8k": (3840, 2160, 7680, 4320),        # configs[4]
}


def test_checker_is_the_reference_on_the_gpu_box(checker):
    """Informational guard: states which checker the whole-frame tests used (it must be `_ref` whenever it travelled)."""
    import cpu_oracle
    assert checker.kind == ("reference" if cpu_oracle.have_ref() else "port")


def test_config0_540p_to_1080p_rgba32f_easu_only(fsr, checker):
    """BASELINE configs[0]: fp32 FsrEasuF, EASU only, RGBA32F storage (RWTexture2D<float4>, FSR_Pass.glsl:40)."""
    iw, ih, ow, oh = SHAPES["540p_to_1080p"]
    img = frames.synthetic_frame(iw, ih, k=0, dtype=np.float32)
    con = checker.FsrEasuCon(iw, ih, iw, ih, ow, oh)
    want = checker.easu_f(img, ow, oh, con)
    src = dev(img)
    out = torch.zeros(oh, ow, 4, dtype=torch.float32, device="cuda")
    fsr.easu(src, out, con=con, flags=fsr.FLAG_MATH_EXACT)
    assert_exact32(host(out), want, "540p->1080p RGBA32F EASU EXACT")
    out.zero_()
    fsr.easu(src, out, con=con, flags=fsr.FLAG_MATH_EXACT | fsr.FLAG_NO_FAST_PATHS)
    assert_exact32(host(out), want, "540p->1080p RGBA32F EASU EXACT (generic kernel)")
    out.zero_()
    fsr.easu(src, out, con=con)
    assert_f_class(host(out), want, "540p->1080p RGBA32F EASU F")
    # the EASU-only HDR mode squares the result (Sample.x = hdr && !useRcas, FSR_Filter.cpp:107, FSR_Pass.hlsl:78-79)
    want_hdr = checker.easu_f(img, ow, oh, con, 4)
    out.zero_()
    fsr.easu(src, out, con=con, flags=fsr.FLAG_MATH_EXACT | fsr.FLAG_HDR_SQUARE)
    assert_exact32(host(out), want_hdr, "540p->1080p RGBA32F EASU EXACT hdr")


@pytest.mark.parametrize("name", ["1080p_to_4k", "1440p_to_4k", "4k_to_8k", "540p_to_1080p"])
def test_whole_frame_two_pass_and_fused(fsr, checker, name):
    """EASU + RCAS on one whole frame, RGBA16F: EXACT = 0 differing values, F <= 1 ULP, fused == two-pass."""
    iw, ih, ow, oh = SHAPES[name]
    img = frames.synthetic_frame(iw, ih, k=1, dtype=np.float16)
    img32 = img.astype(np.float32)
    con = checker.FsrEasuCon(iw, ih, iw, ih, ow, oh)
    rc = checker.FsrRcasCon(0.25)
    want_mid = checker.easu_f(img32, ow, oh, con)
    src = dev(img)
    mid = torch.zeros(oh, ow, 4, dtype=torch.float16, device="cuda")
    out = torch.zeros_like(mid)
    fus = torch.zeros_like(mid)
    for exact in (True, False):
        fl = fsr.FLAG_MATH_EXACT if exact else 0
        tag = "%s %s" % (name, "EXACT" if exact else "F")
        mid.zero_(); out.zero_(); fus.zero_()
        fsr.easu(src, mid, con=con, flags=fl)
        fsr.rcas(mid, out, con=rc, flags=fl)
        fsr.easu_rcas_fused(src, fus, easu_con=con, rcas_con=rc, flags=fl)
        got_mid = host(mid)
        (assert_exact16 if exact else assert_f_class)(got_mid, want_mid, tag + " easu")
        want_out = checker.rcas_f(got_mid.astype(np.float32), rc)  # identical input: the GPU's own intermediary
        (assert_exact16 if exact else assert_f_class)(host(out), want_out, tag + " rcas")
        assert torch.equal(out.view(torch.int16), fus.view(torch.int16)), tag + ": fused launch differs from the two dispatches"
        if name != "4k_to_8k":  # the generic kernels (no exact-2x / fast paths) on the same frame
            mid2 = torch.zeros_like(mid)
            fsr.easu(src, mid2, con=con, flags=fl | fsr.FLAG_NO_FAST_PATHS)
            assert torch.equal(mid.view(torch.int16), mid2.view(torch.int16)), tag + ": generic EASU kernel differs from the fast path"


@pytest.mark.parametrize("name", ["1080p_to_4k", "1440p_to_4k", "4k_to_8k"])
def test_whole_frame_packed_fp16(fsr, checker, name):
    """The packed-binary16 path (the reference's shipping default, FSR_Pass.hlsl:81-87): bit-exact vs FsrEasuH / FsrRcasH."""
    iw, ih, ow, oh = SHAPES[name]
    img = frames.synthetic_frame(iw, ih, k=2, dtype=np.float16)
    con = checker.FsrEasuCon(iw, ih, iw, ih, ow, oh)
    rc = checker.FsrRcasCon(0.25)
    src = dev(img)
    mid = torch.zeros(oh, ow, 4, dtype=torch.float16, device="cuda")
    out = torch.zeros_like(mid)
    fsr.easu(src, mid, con=con, flags=fsr.FLAG_MATH_PACKED_FP16)
    fsr.rcas(mid, out, con=rc, flags=fsr.FLAG_MATH_PACKED_FP16)
    got_mid = host(mid)
    assert_exact16(got_mid, checker.easu_h(img.astype(np.float32), ow, oh, con), name + " easu H")
    assert_exact16(host(out), checker.rcas_h(got_mid.astype(np.float32), rc), name + " rcas H")
    if getattr(checker, "has_hx2", False) and name == "1080p_to_4k":
        # the packed two-pixel entry point compiled from the reference (FsrRcasHx2 + FsrRcasDepackHx2, ffx_fsr1.h:880-984): a whole 4K frame
        assert_exact16(host(out), checker.rcas_hx2(got_mid.astype(np.float32), rc), name + " rcas Hx2")
    fus = torch.zeros_like(mid)
    fsr.easu_rcas_fused(src, fus, easu_con=con, rcas_con=rc, flags=fsr.FLAG_MATH_PACKED_FP16)
    assert torch.equal(out.view(torch.int16), fus.view(torch.int16)), name + ": fused H launch differs from the two H dispatches"


@pytest.mark.parametrize("name", ["1080p_to_4k", "1440p_to_4k"])
def test_whole_frame_rgba8(fsr, checker, name):
    """RGBA8 UNORM storage at full size: EXACT reproduces every code of the oracle run on the decoded image."""
    from test_gpu_unorm import decode, encode, rgba8_frame, check_codes
    iw, ih, ow, oh = SHAPES[name]
    codes = rgba8_frame(iw, ih, 3)
    con = checker.FsrEasuCon(iw, ih, iw, ih, ow, oh)
    rc = checker.FsrRcasCon(0.25)
    src = torch.from_numpy(codes).cuda()
    for exact in (True, False):
        fl = fsr.FLAG_MATH_EXACT if exact else 0
        mid = torch.zeros(oh, ow, 4, dtype=torch.uint8, device="cuda")
        dst = torch.zeros_like(mid)
        fus = torch.zeros_like(mid)
        fsr.easu(src, mid, con=con, flags=fl)
        fsr.rcas(mid, dst, con=rc, flags=fl)
        fsr.easu_rcas_fused(src, fus, easu_con=con, rcas_con=rc, flags=fl)
        got_mid = host(mid)
        want_mid = encode(checker.easu_f(decode(codes, 255), ow, oh, con), 255)
        want_mid[..., 3] = 255
        check_codes(got_mid, want_mid, exact, "%s rgba8 easu exact=%s" % (name, exact))
        want = encode(checker.rcas_f(decode(got_mid, 255), rc), 255)
        want[..., 3] = 255
        check_codes(host(dst), want, exact, "%s rgba8 rcas exact=%s" % (name, exact))
        assert torch.equal(dst, fus), "%s rgba8: fused differs from two-pass" % name


def test_whole_frame_rgb10a2(fsr, checker):
    """R10G10B10A2 UNORM — the format the sample's HDR path renders into and presents (sample/src/DX12/FSR_Filter.cpp:72-73,
    SampleRenderer.cpp:193) — at full size, 1080p -> 4K: EXACT reproduces every 10-bit (and 2-bit alpha) code of the
    reference run on the decoded image, the default arithmetic is within one code with >= 99.5 % equal; the fused launch
    equals the two dispatches word for word."""
    from test_gpu_unorm import pack10, unpack10, check_codes
    iw, ih, ow, oh = SHAPES["1080p_to_4k"]
    img = frames.synthetic_frame(iw, ih, k=4, dtype=np.float32)
    img[..., 3] = (np.arange(iw)[None, :] % 4) / 3.0
    words = pack10(img)
    con = checker.FsrEasuCon(iw, ih, iw, ih, ow, oh)
    rc = checker.FsrRcasCon(0.25)
    src = torch.from_numpy(words.view(np.int32)).cuda()
    fields = ((0, 0x3FF, "R"), (10, 0x3FF, "G"), (20, 0x3FF, "B"), (30, 0x3, "A"))
    want_mid = pack10(checker.easu_f(unpack10(words), ow, oh, con))  # alpha = 1 out of EASU
    for exact in (True, False):
        fl = fsr.FLAG_MATH_EXACT if exact else 0
        mid = torch.zeros(oh, ow, dtype=torch.int32, device="cuda")
        dst = torch.zeros_like(mid)
        fus = torch.zeros_like(mid)
        fsr.easu(src, mid, con=con, flags=fl)
        fsr.rcas(mid, dst, con=rc, flags=fl | fsr.FLAG_RCAS_PASSTHROUGH_ALPHA)
        fsr.easu_rcas_fused(src, fus, easu_con=con, rcas_con=rc, flags=fl | fsr.FLAG_RCAS_PASSTHROUGH_ALPHA)
        got_mid = host(mid).view(np.uint32)
        for sh, n, ch in fields:
            check_codes((got_mid >> sh) & n, (want_mid >> sh) & n, exact, "rgb10a2 easu %s exact=%s" % (ch, exact))
        want = pack10(checker.rcas_f(unpack10(got_mid), rc, 2))  # 2 = pass alpha through; judged on the GPU's own intermediary
        got = host(dst).view(np.uint32)
        for sh, n, ch in fields:
            check_codes((got >> sh) & n, (want >> sh) & n, exact, "rgb10a2 rcas %s exact=%s" % (ch, exact))
        assert torch.equal(dst, fus), "rgb10a2: fused differs from two-pass"


def _batch(fsr, checker, name, n, pipelines):
    iw, ih, ow, oh = SHAPES[name]
    base = [dev(frames.synthetic_frame(iw, ih, k=k, dtype=np.float16)) for k in range(2)]
    src = torch.stack([torch.roll(base[f % 2], shifts=(3 * f, 5 * f), dims=(0, 1)) for f in range(n)]).contiguous()
    con = checker.FsrEasuCon(iw, ih, iw, ih, ow, oh)
    rc = checker.FsrRcasCon(0.25)
    mids, outs = {}, {}
    for exact in (True, False):
        fl = fsr.FLAG_MATH_EXACT if exact else 0
        tag = "%s x%d %s" % (name, n, "EXACT" if exact else "F")
        mids[exact] = torch.zeros(n, oh, ow, 4, dtype=torch.float16, device="cuda")
        outs[exact] = torch.zeros_like(mids[exact])
        fsr.easu(src, mids[exact], con=con, flags=fl)      # one launch over the whole batch
        fsr.rcas(mids[exact], outs[exact], con=rc, flags=fl)
        if "fused" in pipelines:
            fus = torch.zeros_like(mids[exact])
            fsr.easu_rcas_fused(src, fus, easu_con=con, rcas_con=rc, flags=fl)
            assert torch.equal(outs[exact].view(torch.int16), fus.view(torch.int16)), tag + ": fused batch differs from the two-pass batch"
            del fus
    for f in range(n):  # every frame of the batch, whole; the reference's EASU is evaluated once per frame, compared on the device
        want_mid = checker.easu_f(host(src[f]).astype(np.float32), ow, oh, con)
        for exact in (True, False):
            tag = "%s x%d %s frame %d" % (name, n, "EXACT" if exact else "F", f)
            (gpu_assert_exact16 if exact else gpu_assert_f_class)(mids[exact][f], want_mid, tag + " easu")
            if exact or f % 4 == 0:  # RCAS is judged on the GPU's own intermediary: EXACT on every frame, F on every fourth
                want = checker.rcas_f(host(mids[exact][f]).astype(np.float32), rc)
                (gpu_assert_exact16 if exact else gpu_assert_f_class)(outs[exact][f], want, tag + " rcas")


def test_batch_1440p_to_4k_x8_all_frames(fsr, checker):
    """BASELINE configs[2]'s per-GPU shard: 8 frames of 2560x1440 -> 3840x2160 in one launch, every frame checked whole."""
    _batch(fsr, checker, "1440p_to_4k", 8, ("two-pass", "fused"))


def test_batch_4k_to_8k_x16_all_frames(fsr, checker):
    """BASELINE configs[4]'s per-GPU shard: 16 frames of 3840x2160 -> 7680x4320 in one launch, every frame checked whole."""
    _batch(fsr, checker, "4k_to_8k", 16, ("two-pass", "fused"))


def test_dynamic_resolution_whole_frame(fsr, checker):
    """FsrEasuConOffset (ffx_fsr1.h:205-225): a 1600x900 viewport at offset (160, 90) inside a 1920x1080 resource,
    upscaled to 3840x2160 — taps near the viewport's border read real texels of the resource, not clamped ones, and the
    clamp happens at the RESOURCE edge (ffx_fsr1.h:161-166).  Whole frame, EXACT = 0 differing values, default <= 1 ULP."""
    rw, rh, vw, vh, offx, offy, ow, oh = 1920, 1080, 1600, 900, 160.0, 90.0, 3840, 2160
    img = frames.synthetic_frame(rw, rh, k=5, dtype=np.float16)
    con = checker.FsrEasuConOffset(vw, vh, rw, rh, ow, oh, offx, offy)
    assert np.array_equal(con, fsr.FsrEasuConOffset(vw, vh, rw, rh, ow, oh, offx, offy))
    want = checker.easu_f(img.astype(np.float32), ow, oh, con)
    src = dev(img)
    out = torch.zeros(oh, ow, 4, dtype=torch.float16, device="cuda")
    fsr.easu(src, out, con=con, flags=fsr.FLAG_MATH_EXACT)
    assert_exact16(host(out), want, "dynamic resolution EXACT")
    out.zero_()
    fsr.easu(src, out, con=con)
    assert_f_class(host(out), want, "dynamic resolution F")
    out.zero_()
    fsr.easu(src, out, con=con, flags=fsr.FLAG_MATH_PACKED_FP16)
    assert_exact16(host(out), checker.easu_h(img.astype(np.float32), ow, oh, con), "dynamic resolution H")
