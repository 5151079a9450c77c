"""The oracle itself: the plain-C restatement (oracle/fsr1_oracle.c) against the committed golden
vectors (generated from the reference headers compiled verbatim, tests/golden/gen_golden.py), and —
where oracle/_ref is available — against that build directly on fresh inputs."""
import importlib

import numpy as np
import pytest

from conftest import PIXEL_CASES, load_golden, same_bits

frames = importlib.import_module("fidelityfx-fsr_amd.frames")


@pytest.mark.parametrize("name", PIXEL_CASES)
def test_port_matches_golden(port, name):
    g = load_golden(name)
    img = g["input"].astype(np.float32)
    oh, ow = g["easu_f"].shape[:2]
    assert same_bits(port.easu_f(img, ow, oh, g["con"]), g["easu_f"])
    assert same_bits(port.easu_f(img, ow, oh, g["con"], 4), g["easu_f_hdr"])
    assert same_bits(port.easu_h(img, ow, oh, g["con"]), g["easu_h"].astype(np.float32))
    mid = g["mid"].astype(np.float32)
    for fl in range(4):
        assert same_bits(port.rcas_f(mid, g["rcas_con"], fl), g["rcas_f_%d" % fl]), fl
        assert same_bits(port.rcas_h(mid, g["rcas_con"], fl), g["rcas_h_%d" % fl].astype(np.float32)), fl
    assert same_bits(port.rcas_f(mid, g["rcas_con"], 4), g["rcas_f_hdr"])


def test_port_matches_survey_kat_b2(port):
    """pixel values printed from the reference headers in the survey session."""
    g = load_golden("kat_b2")
    assert np.array_equal(frames.kat_frame_64x36(), g["input"])
    e = port.easu_f(g["input"], 128, 72, g["con"])
    r = port.rcas_f(e, port.FsrRcasCon(0.25))
    assert same_bits(e, g["easu_f"]) and same_bits(r, g["rcas_f"])
    kat = {(0, 0): (0.402040273, 0.800000012, 0.256619126, 0.400772572, 0.797477484, 0.255809963),
           (17, 9): (0.475677282, 0.800000012, 0.452284902, 0.507699788, 0.842310548, 0.471487015),
           (127, 71): (0.488122463, 0.800000012, 0.427852303, 0.486583352, 0.797477484, 0.426503211),
           (64, 36): (0.462142944, 0.800000012, 0.366188824, 0.465003610, 0.798840940, 0.339659005)}
    for (x, y), v in kat.items():
        np.testing.assert_allclose(e[y, x, :3], v[:3], rtol=1e-6)
        np.testing.assert_allclose(r[y, x, :3], v[3:], rtol=1e-6)
    assert abs(r.astype(np.float64).sum() - 19292.897310251) < 1e-6
    assert not np.isnan(r).any()


@pytest.mark.parametrize("shape", [(61, 35, 122, 70), (53, 31, 69, 41), (45, 28, 77, 48), (32, 32, 48, 48), (9, 7, 18, 14), (3, 2, 11, 9)])
def test_port_matches_reference_build(port, ref, shape):
    iw, ih, ow, oh = shape
    img = frames.synthetic_frame(iw, ih, k=5, dtype=np.float32)
    con = ref.FsrEasuCon(iw, ih, iw, ih, ow, oh)
    assert np.array_equal(con, port.FsrEasuCon(iw, ih, iw, ih, ow, oh))
    ef = ref.easu_f(img, ow, oh, con)
    assert same_bits(port.easu_f(img, ow, oh, con), ef)
    assert same_bits(port.easu_h(img, ow, oh, con), ref.easu_h(img, ow, oh, con))
    mid = ef.astype(np.float16).astype(np.float32)
    for stops in (0.0, 0.25, 2.0):
        rc = ref.FsrRcasCon(stops)
        for fl in (0, 1, 2, 3, 4):
            assert same_bits(port.rcas_f(mid, rc, fl), ref.rcas_f(mid, rc, fl)), (stops, fl)
            assert same_bits(port.rcas_h(mid, rc, fl), ref.rcas_h(mid, rc, fl)), (stops, fl)


@pytest.mark.parametrize("shape", [(131, 83), (64, 16), (23, 9), (16, 5), (9, 4), (8, 3), (1, 1)], ids=lambda s: "%dx%d" % s)
def test_reference_hx2_is_h_lane_by_lane(port, ref, shape):
    """FsrRcasHx2 (ffx_fsr1.h:888-984, two pixels eight columns apart in the halves of every operand) compiled from the
    reference and compared with FsrRcasH (:782-866) of the same build, pixel for pixel: the packed form runs the same
    binary16 operations per lane.  That is what lets one H kernel (csrc/fsr1_rcas_h.hip) stand for both entry points —
    pinned here by the compiled token stream, not by reading.  Widths that are not multiples of 16 included."""
    w, h = shape
    img = frames.synthetic_frame(w, h, k=7, dtype=np.float16).astype(np.float32)
    img[..., 3] = (np.arange(w, dtype=np.float32)[None, :] / max(w, 1)).astype(np.float16)  # alpha worth passing through
    hostile = frames.adversarial_frame(w, h, k=1).astype(np.float32) if hasattr(frames, "adversarial_frame") else img
    for src in (img, hostile):
        for stops in (0.0, 0.25, 2.0):
            rc = ref.FsrRcasCon(stops)
            for fl in range(8):
                a, b = ref.rcas_hx2(src, rc, fl), ref.rcas_h(src, rc, fl)
                assert same_bits(a, b), (shape, stops, fl)
                assert same_bits(port.rcas_h(src, rc, fl), a), (shape, stops, fl)  # and the restatement equals both


@pytest.mark.parametrize("name", PIXEL_CASES)
def test_reference_hx2_matches_golden(ref, name):
    g = load_golden(name)
    mid = g["mid"].astype(np.float32)
    for fl in range(4):
        assert same_bits(ref.rcas_hx2(mid, g["rcas_con"], fl), g["rcas_h_%d" % fl].astype(np.float32)), fl


def test_port_dynamic_resolution_viewport(port, ref):
    """Viewport smaller than the resource + offset (FsrEasuConOffset): taps clamp at the resource edge."""
    img = frames.synthetic_frame(64, 48, k=2, dtype=np.float32)
    con = ref.FsrEasuConOffset(40, 30, 64, 48, 80, 60, 8, 6)
    assert np.array_equal(con, port.FsrEasuConOffset(40, 30, 64, 48, 80, 60, 8, 6))
    assert same_bits(port.easu_f(img, 80, 60, con), ref.easu_f(img, 80, 60, con))


def test_rcas_black_pixels_nan_policy(port):
    """ffx_fsr1.h:750: mx4 = 0 -> rcp(0)=inf -> 0*inf = NaN -> max() must drop the NaN.
    An all-black frame comes out all zero; a pure-green area keeps finite values."""
    z = np.zeros((8, 8, 4), np.float32)
    z[..., 3] = 1
    out = port.rcas_f(z, port.FsrRcasCon(0.0))
    assert not np.isnan(out).any() and np.all(out[..., :3] == 0)
    g = z.copy()
    g[..., 1] = 0.5
    g[3, 3, 1] = 0.8
    out = port.rcas_f(g, port.FsrRcasCon(0.0))
    assert not np.isnan(out).any()
    assert np.all(out[..., 0] == 0) and np.all(out[..., 2] == 0)


def test_armp8x8(port, ref):
    """ffx_a.h:2304 lane remap: a bijection onto the 8x8 tile keeping 2x2 quads together."""
    import ctypes
    seen = set()
    for lane in range(64):
        a = (ctypes.c_uint32 * 2)()
        b = (ctypes.c_uint32 * 2)()
        port.lib.oracle_ARmp8x8(ctypes.c_uint32(lane), a)
        ref.lib.ref_rmp8x8(ctypes.c_uint32(lane), b)
        assert tuple(a) == tuple(b)
        seen.add(tuple(a))
    assert len(seen) == 64 and all(0 <= x < 8 and 0 <= y < 8 for x, y in seen)
