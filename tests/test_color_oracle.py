"""Colour stages around the filters — FsrLfgaF, FsrSrtmF / FsrSrtmInvF, FsrTepdDitF, FsrTepdC8F / C10F
(ffx-fsr/ffx_fsr1.h:986-1199): the plain-C restatement against the committed golden vectors (generated from the
reference headers compiled verbatim), against that build on fresh inputs, and the properties the header states."""
import importlib

import numpy as np
import pytest

from conftest import load_golden, same_bits

frames = importlib.import_module("fidelityfx-fsr_amd.frames")
PARAMS = dict(amount=0.75, bias=0.0, frame=3, noise_offset=(5, -3))  # tests/golden/gen_golden.py COLOR_PARAMS


def golden_cases(g):
    for k in sorted(g):
        if k.startswith("out_"):
            _, st, key = k.split("_")
            yield int(st), key, g[k]


def test_port_matches_golden_color(port):
    g = load_golden("color_stages")
    noise = g["noise"].astype(np.float32)
    n = 0
    for st, key, want in golden_cases(g):
        got = port.color_f(g[key].astype(np.float32), st, noise=noise, **PARAMS)
        assert same_bits(got, want), (st, key)
        n += 1
    assert n == 12
    dit = np.array([[port.FsrTepdDitF(x, y, f) for x in (0, 1, 17, 3839, 7679)] for y, f in ((0, 0), (5, 1), (2159, 7), (4319, 1000))], np.float32)
    assert same_bits(dit, g["dit"])


def test_port_matches_golden_color_h(port):
    """Half-precision entry points (FsrSrtmH, FsrLfgaH, FsrSrtmInvH, FsrTepdC8H / C10H): restatement vs the committed vectors."""
    g = load_golden("color_stages")
    noise = g["noise"].astype(np.float32)
    n = 0
    for k in sorted(g):
        if k.startswith("outh_"):
            _, st, key = k.split("_")
            got = port.color_h(g[key].astype(np.float32), int(st), noise=noise, **PARAMS)
            assert same_bits(got, g[k].astype(np.float32)), (st, key)
            n += 1
    assert n == 12


@pytest.mark.parametrize("stages", [1, 2, 4, 8, 16, 8 | 32, 2 | 8, 1 | 2 | 4, 2 | 16 | 32])
def test_port_matches_reference_build_color_h(port, ref, stages):
    rng = np.random.default_rng(50 + stages)
    img = rng.random((29, 43, 4)).astype(np.float32)
    if stages & 1:
        img[..., :3] = img[..., :3] ** 4 * 30000.0
    img = img.astype(np.float16).astype(np.float32)
    noise = (rng.random((2, 5, 7, 4)).astype(np.float32) - np.array([0.5, 0.5, 0.5, 0.0], np.float32)).astype(np.float16).astype(np.float32)
    for frame in (0, 3, 77777):
        kw = dict(amount=0.4, bias=0.1, frame=frame, noise=noise, noise_offset=(-11, 13))
        assert same_bits(ref.color_h(img, stages, **kw), port.color_h(img, stages, **kw)), frame


@pytest.mark.parametrize("stages", [1, 2, 4, 8, 16, 8 | 32, 2 | 8, 2 | 4, 1 | 2 | 4, 1 | 2 | 16, 2 | 16 | 32])
def test_port_matches_reference_build_color(port, ref, stages):
    rng = np.random.default_rng(stages)
    img = rng.random((33, 47, 4)).astype(np.float32)
    if stages & 1:
        img[..., :3] = (img[..., :3] ** 4 * 30000.0)
    img[0, :6, :3] = np.array([0.0, 1.0, 65504.0 if stages & 1 else 0.5, 1e-30, 0.25, 1.0 - 2.0 ** -24], np.float32)[:, None]
    noise = rng.random((3, 5, 7, 4)).astype(np.float32)
    noise[..., :3] -= 0.5
    for frame in (0, 1, 2, 4000000000):
        a = ref.color_f(img, stages, amount=0.4, bias=0.1, frame=frame, noise=noise, noise_offset=(-11, 13))
        b = port.color_f(img, stages, amount=0.4, bias=0.1, frame=frame, noise=noise, noise_offset=(-11, 13))
        assert same_bits(a, b), frame


def test_srtm_round_trip_and_range(port):
    """FsrSrtmF maps {0..FP16_MAX} into {0..1} preserving the RGB ratio; FsrSrtmInvF undoes it (:1036-1040)."""
    rng = np.random.default_rng(7)
    hdr = (rng.random((16, 16, 4)) ** 6 * 1000.0).astype(np.float32)
    t = port.color_f(hdr, 1)
    assert t[..., :3].min() >= 0.0 and t[..., :3].max() < 1.0
    back = port.color_f(t, 4)
    # 1 - max3(c) cancels: a peak of 1000 leaves 1e-3, i.e. ~2^-14 relative; beyond 32768 the inverse saturates (:1044)
    np.testing.assert_allclose(back[..., :3], hdr[..., :3], rtol=5e-4, atol=1e-6)
    big = port.color_f(port.color_f(np.full((1, 1, 4), 60000.0, np.float32), 1), 4)
    assert 16384.0 <= big[0, 0, 0] <= 32768.0
    ratio_in = hdr[..., 0] / np.maximum(hdr[..., 1], 1e-20)
    ratio_out = t[..., 0] / np.maximum(t[..., 1], 1e-20)
    np.testing.assert_allclose(ratio_out, ratio_in, rtol=1e-5)
    # the c = 1.0 case the header calls out (:1043): no division by zero, peak 32768
    one = np.ones((1, 1, 4), np.float32)
    assert port.color_f(one, 4)[0, 0, 0] == 32768.0


def test_lfga_limits_grain_by_distance_to_signal_limits(port):
    """:1004 'These functions limit grain based on distance to signal limits': 0 and 1 never move, the result stays in [0,1]."""
    rng = np.random.default_rng(3)
    img = rng.random((20, 20, 4)).astype(np.float32)
    img[0, 0, :3] = 0.0
    img[0, 1, :3] = 1.0
    noise = (rng.random((1, 4, 4, 4)) - 0.5).astype(np.float32)
    out = port.color_f(img, 2, amount=1.0, noise=noise)
    assert np.all(out[0, 0, :3] == 0.0) and np.all(out[0, 1, :3] == 1.0)
    assert out[..., :3].min() >= 0.0 and out[..., :3].max() <= 1.0
    assert np.array_equal(out[..., 3], img[..., 3])
    assert same_bits(port.color_f(img, 2, amount=0.0, noise=noise), img)  # amount 0 is the identity


@pytest.mark.parametrize("stage,steps", [(8, 255), (16, 1023)])
def test_tepd_is_on_the_code_grid_and_preserves_energy(port, stage, steps):
    """FsrTepdC8F / C10F output the gamma-2.0 value of one of the two codes around sqrt(c) (:1056-1062), and with an
    unbiased dither the *linear* mean over time equals c ('temporally energy preserving')."""
    c = np.linspace(0.0, 1.0, 97, dtype=np.float32) ** 2
    img = np.zeros((1, 97, 4), np.float32)
    img[0, :, :3] = c[:, None]
    acc = np.zeros(97, np.float64)
    n = 256
    for i in range(n):
        dit = np.full((1, 1, 1, 4), (i + 0.5) / n, np.float32)
        out = port.color_f(img, stage | 32, noise=dit)[0, :, 0]
        codes = out.astype(np.float64) * steps
        assert np.all(np.abs(codes - np.round(codes)) < 1e-3)
        lo = np.floor(np.sqrt(c.astype(np.float64)) * steps + 1e-4)
        assert np.all((np.round(codes) >= lo - 1) & (np.round(codes) <= lo + 1))
        acc += (np.round(codes) / steps) ** 2
    np.testing.assert_allclose(acc / n, c, atol=2.5 / n / steps + 1e-6)


def test_tepd_dither_pattern(port):
    """FsrTepdDitF (:1082-1091) is fract((x + f) * phi + y / 3.69): in [0, 1), shifted by the frame index."""
    v = np.array([[port.FsrTepdDitF(x, y, 0) for x in range(64)] for y in range(16)])
    assert v.min() >= 0.0 and v.max() < 1.0
    assert port.FsrTepdDitF(5, 3, 7) == port.FsrTepdDitF(12, 3, 0)
    phi = (1.0 + 5.0 ** 0.5) / 2.0
    want = np.array([[(x * phi + y / 3.69) % 1.0 for x in range(64)] for y in range(16)])
    d = np.abs(v - want)
    assert np.all(np.minimum(d, 1.0 - d) < 2e-5)
