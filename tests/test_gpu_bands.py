"""Row bands of one frame (a single frame split over several GPUs with zero exchange — every GPU holds the
input, runs EASU on its band plus one row either side and RCAS on the band).  Here the bands are computed one after the
other on one GPU and must reassemble into the full-frame result BIT FOR BIT, for both arithmetics: the position arithmetic
of ffx_fsr1.h:324-326 runs on full-image coordinates whatever the band, and the rows RCAS reads across a band boundary are
real EASU rows, not the zeros of the image border (FSR_Pass.hlsl:61)."""
import importlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
frames = importlib.import_module("fidelityfx-fsr_amd.frames")


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def splits(total, n, even):
    """n bands covering [0, total): boundaries rounded to even rows when `even` (the exact-2x kernel's quads), arbitrary otherwise"""
    cuts = [0]
    for k in range(1, n):
        c = total * k // n + (0 if even else 1)
        c = c // 2 * 2 if even else c | 1
        cuts.append(min(max(c, cuts[-1] + 1), total - 1))
    return list(zip(cuts, cuts[1:] + [total]))


@pytest.mark.parametrize("shape", [(960, 540, 1920, 1080), (640, 360, 960, 540), (97, 61, 131, 83), (64, 40, 128, 80)],
                         ids=lambda s: "%dx%d_to_%dx%d" % s)
@pytest.mark.parametrize("n_bands", [2, 3, 5])
def test_bands_reassemble_into_the_full_frame(fsr, shape, n_bands):
    iw, ih, ow, oh = shape
    src = dev(frames.synthetic_frame(iw, ih, k=8, dtype=np.float16))
    for flags in (0, fsr.FLAG_MATH_EXACT):
        mid = torch.zeros(oh, ow, 4, dtype=torch.float16, device="cuda")
        full = torch.zeros_like(mid)
        fsr.easu(src, mid, flags=flags)
        fsr.rcas(mid, full, sharpness=0.25, flags=flags)
        for even in (True, False):
            out = torch.full_like(full, -1.0)
            for (y0, y1) in splits(oh, n_bands, even):
                fsr.upscale_band(src, out[y0:y1], (ow, oh), (y0, y1), sharpness=0.25, flags=flags)
            assert torch.equal(out.view(torch.int16), full.view(torch.int16)), "bands %s (flags %d, even %s) differ from the full frame" % (
                splits(oh, n_bands, even), flags, even)


@pytest.mark.parametrize("shape", [(960, 540, 1920, 1080), (640, 360, 960, 540), (97, 61, 131, 83), (64, 40, 128, 80), (150, 90, 195, 117)],
                         ids=lambda s: "%dx%d_to_%dx%d" % s)
@pytest.mark.parametrize("n_bands", [2, 3, 7])
def test_fused_bands_reassemble_into_the_full_frame(fsr, shape, n_bands):
    """The single-launch band form: the tile aprons compute the rows beyond a band boundary with EASU on full-image
    coordinates, so the bands equal the full-frame fused launch — and hence the two dispatches — bit for bit."""
    iw, ih, ow, oh = shape
    src = dev(frames.synthetic_frame(iw, ih, k=11, dtype=np.float16))
    for flags in (0, fsr.FLAG_MATH_EXACT, fsr.FLAG_RCAS_DENOISE):
        full = torch.zeros(oh, ow, 4, dtype=torch.float16, device="cuda")
        fsr.easu_rcas_fused(src, full, sharpness=0.25, flags=flags)
        for even in (True, False):
            out = torch.full_like(full, -1.0)
            for (y0, y1) in splits(oh, n_bands, even):
                fsr.upscale_band(src, out[y0:y1], (ow, oh), (y0, y1), sharpness=0.25, flags=flags, fused=True)
            assert torch.equal(out.view(torch.int16), full.view(torch.int16)), "fused bands %s (flags %d, even %s) differ from the full frame" % (
                splits(oh, n_bands, even), flags, even)
    two = torch.zeros_like(full)
    for (y0, y1) in splits(oh, n_bands, False):
        fsr.upscale_band(src, two[y0:y1], (ow, oh), (y0, y1), sharpness=0.25, flags=fsr.FLAG_RCAS_DENOISE)
    assert torch.equal(two.view(torch.int16), full.view(torch.int16))  # two-dispatch bands == fused full frame


@pytest.mark.parametrize("shape", [(640, 360, 1280, 720), (97, 61, 131, 83)], ids=lambda s: "%dx%d_to_%dx%d" % s)
def test_hdr_bands_square_once(fsr, shape):
    """FSR1_FLAG_HDR_SQUARE (`c *= c`, FSR_Pass.hlsl:78-79) with RCAS on belongs to the RCAS dispatch only (FSR_Filter.cpp:107:
    EASU's Sample.x is 0 when RCAS follows): the two-dispatch bands, the fused bands and the full frame agree bit for bit,
    and they are what fsr1_upscale(hdr = 1) writes."""
    iw, ih, ow, oh = shape
    src = dev(frames.synthetic_frame(iw, ih, k=5, dtype=np.float16))
    for math in (0, fsr.FLAG_MATH_EXACT):
        flags = math | fsr.FLAG_HDR_SQUARE | fsr.FLAG_RCAS_DENOISE
        mid = torch.zeros(oh, ow, 4, dtype=torch.float16, device="cuda")
        full = torch.zeros_like(mid)
        fsr.easu(src, mid, flags=math)          # Sample.x = 0
        fsr.rcas(mid, full, sharpness=0.25, flags=flags)  # Sample.x = hdr
        fused_full = torch.zeros_like(full)
        fsr.easu_rcas_fused(src, fused_full, sharpness=0.25, flags=flags)
        assert torch.equal(fused_full.view(torch.int16), full.view(torch.int16))
        for fused in (False, True):
            out = torch.full_like(full, -1.0)
            for (y0, y1) in splits(oh, 3, False):
                fsr.upscale_band(src, out[y0:y1], (ow, oh), (y0, y1), sharpness=0.25, flags=flags, fused=fused)
            assert torch.equal(out.view(torch.int16), full.view(torch.int16)), "hdr bands (fused=%s, math=%d) differ from the full frame" % (fused, math)


def test_fused_band_batch_and_rgba8(fsr):
    """Several frames per launch share the band geometry; UNORM storage goes through the same apron logic."""
    iw, ih, ow, oh = 96, 54, 192, 108
    src = torch.stack([dev(frames.synthetic_frame(iw, ih, k=k, dtype=np.float16)) for k in range(3)])
    full = torch.zeros(3, oh, ow, 4, dtype=torch.float16, device="cuda")
    fsr.easu_rcas_fused(src, full, sharpness=0.4)
    con = fsr.FsrEasuCon(iw, ih, iw, ih, ow, oh)
    out = torch.full_like(full, -1.0)
    for (y0, y1) in ((0, 37), (37, 90), (90, 108)):
        fsr.easu_rcas_fused_band(src, out[:, y0:y1], con, sharpness=0.4, origin_y=y0, rows_above=int(y0 > 0), rows_below=int(y1 < oh))
    assert torch.equal(out.view(torch.int16), full.view(torch.int16))
    src8 = (src[0].float().clamp(0, 1) * 255 + 0.5).floor().to(torch.uint8)
    full8 = torch.zeros(oh, ow, 4, dtype=torch.uint8, device="cuda")
    fsr.easu_rcas_fused(src8, full8, sharpness=0.4, flags=fsr.FLAG_MATH_EXACT)
    out8 = torch.zeros_like(full8)
    for (y0, y1) in ((0, 50), (50, 51), (51, 108)):
        fsr.easu_rcas_fused_band(src8, out8[y0:y1], con, sharpness=0.4, origin_y=y0, rows_above=int(y0 > 0), rows_below=int(y1 < oh), flags=fsr.FLAG_MATH_EXACT)
    assert torch.equal(out8, full8)


def test_easu_band_window_with_x_origin(fsr):
    """A window with an x origin as well (a tile of the output, not just a row band), RGBA32F: equals the same window of the full EASU."""
    iw, ih, ow, oh = 200, 120, 300, 180
    src = dev(frames.synthetic_frame(iw, ih, k=3, dtype=np.float32))
    con = fsr.FsrEasuCon(iw, ih, iw, ih, ow, oh)
    full = torch.zeros(oh, ow, 4, dtype=torch.float32, device="cuda")
    fsr.easu(src, full, con=con, flags=fsr.FLAG_MATH_EXACT)
    for (x0, y0, w, h) in ((0, 0, 300, 180), (64, 16, 100, 50), (37, 91, 263, 89), (299, 179, 1, 1)):
        win = torch.zeros(h, w, 4, dtype=torch.float32, device="cuda")
        fsr.easu_band(src, win, con, origin=(x0, y0), flags=fsr.FLAG_MATH_EXACT)
        assert torch.equal(win, full[y0:y0 + h, x0:x0 + w]), (x0, y0, w, h)


def test_upscale_band_with_a_caller_supplied_intermediary(fsr):
    """upscale_band takes a scratch intermediary of at least band rows + 3 (one EASU row either side, one more for the even start);
    a buffer sized to the old "+ 2" contract is refused, never silently truncated (that was an out-of-bounds read)."""
    iw, ih, ow, oh = 64, 40, 128, 80
    src = dev(frames.synthetic_frame(iw, ih, k=5, dtype=np.float16))
    mid_full = torch.zeros(oh, ow, 4, dtype=torch.float16, device="cuda")
    full = torch.zeros_like(mid_full)
    fsr.easu(src, mid_full)
    fsr.rcas(mid_full, full, sharpness=0.25)
    out = torch.full_like(full, -1.0)
    for (y0, y1) in ((0, 26), (26, 54), (54, 80)):  # even cuts: the middle band's intermediary starts two rows above it
        scratch = torch.full((y1 - y0 + 3 + 2, ow, 4), 9.0, dtype=torch.float16, device="cuda")  # two guard rows behind
        fsr.upscale_band(src, out[y0:y1], (ow, oh), (y0, y1), mid=scratch, sharpness=0.25)
        torch.cuda.synchronize()
        assert bool((scratch[-2:] == 9.0).all()), "wrote beyond the rows the band needs"
    assert torch.equal(out.view(torch.int16), full.view(torch.int16))
    with pytest.raises(fsr.Fsr1Error, match="intermediary"):  # rows + 2: too small for an even interior cut
        fsr.upscale_band(src, out[26:54], (ow, oh), (26, 54), mid=torch.zeros(28 + 2, ow, 4, dtype=torch.float16, device="cuda"))
    with pytest.raises(fsr.Fsr1Error, match="intermediary"):  # wrong width
        fsr.upscale_band(src, out[26:54], (ow, oh), (26, 54), mid=torch.zeros(40, ow + 2, 4, dtype=torch.float16, device="cuda"))
    with pytest.raises(fsr.Fsr1Error, match="dtype"):
        fsr.upscale_band(src, out[26:54], (ow, oh), (26, 54), mid=torch.zeros(40, ow, 4, dtype=torch.float32, device="cuda"))


def test_band_argument_validation(fsr):
    src = dev(frames.synthetic_frame(32, 18, k=1, dtype=np.float16))
    band = torch.zeros(10, 64, 4, dtype=torch.float16, device="cuda")
    con = fsr.FsrEasuCon(32, 18, 32, 18, 64, 36)
    with pytest.raises(fsr.Fsr1Error):
        fsr.easu_band(src, band, con, origin=(0, -2))
    with pytest.raises(fsr.Fsr1Error):
        fsr.easu_band(src, band, con, origin=(0, 4), flags=fsr.FLAG_MATH_PACKED_FP16)
    mid = torch.zeros(12, 64, 4, dtype=torch.float16, device="cuda")
    with pytest.raises(fsr.Fsr1Error):
        fsr.rcas_band(mid[1:11], band, rows_above=2, rows_below=0)
    with pytest.raises(fsr.Fsr1Error):  # the row above the view is the output buffer itself: overlap
        fsr.rcas_band(mid[1:11], mid[0:10], rows_above=1, rows_below=0)
    with pytest.raises(fsr.Fsr1Error):
        fsr.easu_rcas_fused_band(src, band, con, origin_y=0, rows_above=1)  # no row above row 0
    with pytest.raises(fsr.Fsr1Error):
        fsr.easu_rcas_fused_band(src, band, con, origin_y=4, rows_above=1, rows_below=3)
    with pytest.raises(fsr.Fsr1Error):
        fsr.easu_rcas_fused_band(src, band, con, origin_y=4, rows_above=1, rows_below=1, flags=fsr.FLAG_MATH_PACKED_FP16)
