"""The contracted twin of the reference build (oracle/build_ref.sh: -ffp-contract=fast -mfma) is a yardstick, not a parity target:
it must be a DIFFERENT arithmetic from the pinned build (else the self-spread table of README.md measures nothing) and still the
reference — per stage within 1 binary16 ULP of the pinned build, like every conformant compilation (tests/ref_self_spread.py
publishes the whole-frame histograms as profiles/r06_ref_self_spread.json)."""
import importlib
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import cpu_oracle  # noqa: E402
import image_parity  # noqa: E402

frames = importlib.import_module("fidelityfx-fsr_amd.frames")

pytestmark = pytest.mark.skipif(not (cpu_oracle.have_ref() and cpu_oracle.have_ref_fma()), reason="needs both reference builds (oracle/build_ref.sh)")


def test_the_twin_is_another_arithmetic_and_still_the_reference():
    pinned, twin = cpu_oracle.ref(), cpu_oracle.ref_fma()
    img = frames.synthetic_frame(480, 270, k=1).astype(np.float32)
    out_p, mid_p = image_parity.reference_chain(pinned, img, 960, 540, 0.25, return_mid=True)
    out_t, mid_t = image_parity.reference_chain(twin, img, 960, 540, 0.25, return_mid=True)
    # constant setup is host code in both: identical
    assert np.array_equal(pinned.FsrEasuCon(480, 270, 480, 270, 960, 540), twin.FsrEasuCon(480, 270, 480, 270, 960, 540))
    # binary32 results differ (contraction happened) ...
    assert not np.array_equal(twin.easu_f(img, 960, 540, pinned.FsrEasuCon(480, 270, 480, 270, 960, 540)),
                              pinned.easu_f(img, 960, 540, pinned.FsrEasuCon(480, 270, 480, 270, 960, 540)))
    # ... by no more than 1 binary16 ULP per stage on identical input
    assert cpu_oracle.half_ulp_diff(mid_t[..., :3], mid_p[..., :3]).max() <= 1
    assert cpu_oracle.half_ulp_diff(twin.rcas_f(mid_p, pinned.FsrRcasCon(0.25))[..., :3], out_p[..., :3]).max() <= 1
    # end to end the chain's second stage amplifies the first stage's 1-ULP differences: the spread the product's default arithmetic shares
    d = cpu_oracle.half_ulp_diff(out_t[..., :3], out_p[..., :3])
    assert (d <= 1).mean() >= 0.9995
