"""Extreme extents: very wide and very tall images, and row pitches that put most rows beyond the 4 GiB mark — the kernels form
addresses as a 64-bit row base plus a 32-bit offset within the row, which these shapes hold to account.
Compared with the CPU oracle: EXACT bit-identical, packed fp16 bit-identical to the H oracle, fused == two dispatches."""
import importlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
frames = importlib.import_module("fidelityfx-fsr_amd.frames")


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def host(t):
    torch.cuda.synchronize()
    return t.detach().cpu().numpy()


def bits_equal(got16, want_f32):
    g = np.asarray(got16).view(np.uint16)
    w = np.asarray(want_f32, np.float32).astype(np.float16).view(np.uint16)
    return int((g != w).sum())


def pipeline_checks(fsr, port, src, iw, ih, ow, oh, out_alloc, what):
    """EASU / RCAS / fused in EXACT and H on `src` (a device tensor or view), outputs allocated by out_alloc()"""
    img = host(src).astype(np.float32)
    con = port.FsrEasuCon(iw, ih, iw, ih, ow, oh)
    rcon = port.FsrRcasCon(0.25)
    want_e = port.easu_f(img, ow, oh, con)
    mid, out, fused = out_alloc(), out_alloc(), out_alloc()
    fsr.easu(src, mid, con=con, flags=fsr.FLAG_MATH_EXACT)
    assert bits_equal(host(mid), want_e) == 0, what + ": easu EXACT"
    fsr.rcas(mid, out, con=rcon, flags=fsr.FLAG_MATH_EXACT)
    want_r = port.rcas_f(host(mid).astype(np.float32), rcon, 0)
    assert bits_equal(host(out), want_r) == 0, what + ": rcas EXACT"
    fsr.easu_rcas_fused(src, fused, easu_con=con, rcas_con=rcon, flags=fsr.FLAG_MATH_EXACT)
    assert torch.equal(fused.view(torch.int16), out.view(torch.int16)), what + ": fused EXACT != two dispatches"
    fsr.easu(src, mid, con=con, flags=fsr.FLAG_MATH_PACKED_FP16)
    assert bits_equal(host(mid), port.easu_h(img, ow, oh, con)) == 0, what + ": easu H"
    fsr.rcas(mid, out, con=rcon, flags=fsr.FLAG_MATH_PACKED_FP16)
    assert bits_equal(host(out), port.rcas_h(host(mid).astype(np.float32), rcon, 0)) == 0, what + ": rcas H"
    fsr.easu(src, mid, con=con)  # default arithmetic: the same image up to class F
    import cpu_oracle
    d = cpu_oracle.half_ulp_diff(host(mid).astype(np.float32), want_e)
    assert d.max() <= 1 and float((d == 0).mean()) >= 0.995, what + ": easu F"


@pytest.mark.parametrize("shape", [(20000, 6, 40000, 12), (20000, 6, 30000, 9), (6, 20000, 12, 40000), (5, 13000, 9, 23400), (33000, 3, 33000, 3)],
                         ids=lambda s: "%dx%d_to_%dx%d" % s)
def test_very_wide_and_very_tall_images(fsr, port, shape):
    iw, ih, ow, oh = shape
    src = dev(frames.synthetic_frame(iw, ih, k=4, blocks=False, dtype=np.float16))
    pipeline_checks(fsr, port, src, iw, ih, ow, oh, lambda: torch.zeros(oh, ow, 4, dtype=torch.float16, device="cuda"), "%dx%d->%dx%d" % shape)


def test_rows_beyond_4_gib(fsr, port):
    """Row pitch 2 MiB: row 2048 of the input and of every output starts 4 GiB into its allocation, the last output row at ~9.4 GiB."""
    iw, ih, ow, oh = 48, 2400, 96, 4800
    pitch_px = 262144  # x 8 B = 2 MiB per row
    free, _ = torch.cuda.mem_get_info()
    need = pitch_px * 8 * (ih + 3 * oh)
    if free < need + (8 << 30):
        pytest.skip("needs %.0f GiB of device memory" % (need / 2**30))
    big_in = torch.zeros(ih, pitch_px, 4, dtype=torch.float16, device="cuda")
    src = big_in[:, :iw]
    src.copy_(dev(frames.synthetic_frame(iw, ih, k=6, blocks=False, dtype=np.float16)))
    assert src.stride(0) * 2 == 2 << 20 and (ih - 1) * (2 << 20) > 1 << 32
    bufs = []

    def out_alloc():
        b = torch.zeros(oh, pitch_px, 4, dtype=torch.float16, device="cuda")
        bufs.append(b)
        return b[:, :ow]
    pipeline_checks(fsr, port, src, iw, ih, ow, oh, out_alloc, "2 MiB pitch")
    for b in bufs:
        assert float(b[:, ow:ow + 64].abs().sum()) == 0.0 and float(b[:, -64:].abs().sum()) == 0.0  # padding untouched
