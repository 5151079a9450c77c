"""Hostile but finite inputs (HDR highlights near the top of binary16, zeros, negative texels, subnormals): the plain-C
restatement against the reference headers compiled verbatim, for every entry point — the oracle has to be pinned on these
before the GPU is held to it (tests/test_gpu_special_values.py).  Inf / NaN that the arithmetic produces from these
inputs (e.g. binary16 overflow in FsrEasuH) must appear at the same places."""
import importlib

import numpy as np
import pytest

from conftest import same_bits

frames = importlib.import_module("fidelityfx-fsr_amd.frames")


def test_adversarial_frame_contents():
    f = frames.adversarial_frame(96, 64, k=1, dtype=np.float32)
    assert np.isfinite(f).all() and f.max() == 65504.0 and f.min() < 0
    h = f.astype(np.float16)
    assert (np.abs(h[..., :3]) < 6.1e-5).any() and (h[..., :3] == 0).any()  # subnormals and zeros present
    assert np.array_equal(h.astype(np.float32), f)  # every value is a binary16 value


@pytest.mark.parametrize("shape", [(61, 35, 122, 70), (53, 31, 69, 41), (40, 24, 52, 31)], ids=lambda s: "%dx%d_to_%dx%d" % s)
def test_port_matches_reference_build_on_adversarial_values(port, ref, shape):
    iw, ih, ow, oh = shape
    con = ref.FsrEasuCon(iw, ih, iw, ih, ow, oh)
    for k in (0, 1):
        img = frames.adversarial_frame(iw, ih, k=k, dtype=np.float32)
        for fl in (0, 4):
            ef = ref.easu_f(img, ow, oh, con, fl)
            assert same_bits(port.easu_f(img, ow, oh, con, fl), ef), ("easu_f", k, fl)
        assert same_bits(port.easu_h(img, ow, oh, con), ref.easu_h(img, ow, oh, con)), ("easu_h", k)
        mid = frames.adversarial_frame(ow, oh, k=k + 7, dtype=np.float32)
        for stops in (0.0, 0.25):
            rc = ref.FsrRcasCon(stops)
            for fl in (0, 1, 2, 3, 4):
                assert same_bits(port.rcas_f(mid, rc, fl), ref.rcas_f(mid, rc, fl)), ("rcas_f", k, stops, fl)
                assert same_bits(port.rcas_h(mid, rc, fl), ref.rcas_h(mid, rc, fl)), ("rcas_h", k, stops, fl)


@pytest.mark.parametrize("stages", [1, 4, 1 | 4, 2, 1 | 2 | 4, 8, 16, 2 | 8])
def test_colour_stages_on_adversarial_values(port, ref, stages):
    """FsrSrtmF / FsrLfgaF / FsrSrtmInvF / FsrTepdC*F (ffx_fsr1.h:986-1199) on the same hostile values, F and H entry points."""
    rng = np.random.default_rng(77)
    noise = (rng.random((2, 8, 12, 4)).astype(np.float32) - np.array([0.5, 0.5, 0.5, 0.0], np.float32)).astype(np.float16).astype(np.float32)
    for k in (0, 1):
        img = frames.adversarial_frame(70, 41, k=k, dtype=np.float32)
        kw = dict(amount=0.6, bias=0.05, frame=5 + k, noise=noise, noise_offset=(-3, 17))
        assert same_bits(port.color_f(img, stages, **kw), ref.color_f(img, stages, **kw)), ("color_f", k)
        assert same_bits(port.color_h(img, stages, **kw), ref.color_h(img, stages, **kw)), ("color_h", k)
