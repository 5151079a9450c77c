"""Image-level parity: the product's pipelines against the reference's two passes CHAINED on the same input.

The per-stage whole-frame tests (test_gpu_fullframe.py) judge RCAS on the GPU's own intermediary.  This file measures the
end-to-end distance instead — what the benchmarked (default, "F") arithmetic actually delivers against

    FsrEasuF (ffx-fsr/ffx_fsr1.h:315-437) -> RTNE binary16 intermediary -> FsrRcasF (:684-769)

as evaluated by the reference headers compiled here (oracle/_ref), on whole frames of every BASELINE shape AND of the
reference's true-ratio presets (sample/src/DX12/FSRSample.h:79-95: 1477x831 -> 1080p, 2259x1270 -> 4K, 2954x1662 -> 4K), for
the two dispatches, the fused launch, `auto`, and frames sent through an fsr1_pipeline.

    EXACT    the whole chain is bit-identical (0 differing binary16 values)
    default  gated on: no NaN, >= 99 % of the R/G/B values within 1 binary16 ULP of the chain; the full histogram
             (0 / 1 / 2 / 3-4 / > 4 ULP, max, fraction bit-equal) is REPORTED — written to gpurun_out/r05_image_parity.json
             (committed as profiles/r05_image_parity.json) and quoted in README.md.  Values beyond 1 ULP are where RCAS's
             limiter (a ratio of small differences) amplifies a 1-ULP difference of the intermediary.
"""
import importlib
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
frames = importlib.import_module("fidelityfx-fsr_amd.frames")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MIN_WITHIN_1ULP = 0.99
REPORT = {}


@pytest.fixture(scope="module")
def parity():
    import image_parity
    return image_parity


@pytest.fixture(scope="module", autouse=True)
def write_report():
    yield
    if not REPORT:
        return
    out_dir = os.path.join(ROOT, "gpurun_out")
    try:
        os.makedirs(out_dir, exist_ok=True)
        with open(os.path.join(out_dir, "r05_image_parity.json"), "w") as f:
            json.dump(REPORT, f, indent=1, sort_keys=True)
    except OSError:
        pass


SHAPE_NAMES = ["540p_to_1080p", "1080p_to_4k", "1440p_to_4k", "4k_to_8k", "831p_to_1080p", "1270p_to_4k", "1662p_to_4k"]


@pytest.mark.parametrize("name", SHAPE_NAMES)
def test_final_image_against_the_reference_chain(fsr, parity, name):
    iw, ih, ow, oh = parity.SHAPES[name]
    o = parity.checker()
    img = frames.synthetic_frame(iw, ih, k=7, dtype=np.float16)
    want, want_mid = parity.reference_chain(o, img.astype(np.float32), ow, oh, 0.25, return_mid=True)
    src = torch.from_numpy(img).cuda()
    con = fsr.FsrEasuCon(iw, ih, iw, ih, ow, oh)
    rc = fsr.FsrRcasCon(0.25)
    mid = torch.zeros(oh, ow, 4, dtype=torch.float16, device="cuda")

    def two_pass(flags):
        dst = torch.zeros_like(mid)
        fsr.easu(src, mid, con=con, flags=flags)
        fsr.rcas(mid, dst, con=rc, flags=flags)
        return dst

    def fused(flags):
        dst = torch.zeros_like(mid)
        fsr.easu_rcas_fused(src, dst, easu_con=con, rcas_con=rc, flags=flags)
        return dst

    def pipelined(flags, mode):
        pipe = fsr.Pipeline(3)
        dsts = [torch.zeros_like(mid) for _ in range(3)]  # three frames in flight, one per slot; the same input: the outputs must agree
        for d in dsts:
            pipe.upscale(src, d, sharpness=0.25, use_rcas=True, fused=mode, flags=flags)
        pipe.synchronize()
        pipe.close()
        for d in dsts[1:]:
            assert torch.equal(d.view(torch.int16), dsts[0].view(torch.int16)), "%s: frames of one pipeline differ" % name
        return dsts[0]

    entry = {"shape": "%dx%d -> %dx%d" % (iw, ih, ow, oh), "checker": o.kind,
             "reference": "FsrEasuF -> RTNE binary16 -> FsrRcasF (ffx_fsr1.h:315-437, :684-769), sharpness 0.25 stops"}
    # EXACT: the chain is bit-identical end to end
    ex = parity.ulp_histogram(two_pass(fsr.FLAG_MATH_EXACT), want)
    entry["exact_two_dispatch"] = ex
    assert ex["max_ulp"] == 0 and ex["nan_in_output"] == 0 and ex["alpha_equal"], "%s EXACT: the final image differs from the reference chain: %s" % (name, ex)
    ex_mid = parity.ulp_histogram(mid, want_mid)
    assert ex_mid["max_ulp"] == 0, "%s EXACT: the intermediary differs from FsrEasuF: %s" % (name, ex_mid)
    # default arithmetic: every way the product can run the frame
    results = {
        "two_dispatch": two_pass(0),
        "fused": fused(0),
        "pipelined_two_dispatch": pipelined(0, 0),
        "pipelined_auto": pipelined(0, 2),
    }
    for how, img_out in results.items():
        h = parity.ulp_histogram(img_out, want)
        entry["default_" + how] = h
        assert h["nan_in_output"] == 0, "%s default %s: NaN in the output" % (name, how)
        assert h["alpha_equal"], "%s default %s: alpha differs" % (name, how)
        assert h["frac_within_1ulp"] >= MIN_WITHIN_1ULP, "%s default %s: only %.4f of the values within 1 ULP of the reference chain (%s)" % (
            name, how, h["frac_within_1ulp"], h)
    # the product's pipelines agree with each other bit for bit (fused == two dispatches == pipelined)
    base = results["two_dispatch"].view(torch.int16)
    for how, img_out in results.items():
        assert torch.equal(img_out.view(torch.int16), base), "%s: default %s differs from the two dispatches" % (name, how)
    # and the intermediary of the default arithmetic against FsrEasuF alone (the per-stage class, for the same frame)
    fsr.easu(src, mid, con=con)
    entry["default_easu_stage"] = parity.ulp_histogram(mid, want_mid)
    assert entry["default_easu_stage"]["max_ulp"] <= 1
    REPORT[name] = entry
    print("\n%s %s" % (name, json.dumps({k: (v if not isinstance(v, dict) else {kk: v[kk] for kk in ("max_ulp", "frac_bit_equal", "frac_within_1ulp", "hist") if kk in v})
                                         for k, v in entry.items()})))


@pytest.mark.parametrize("name", ["831p_to_1080p", "1270p_to_4k", "1662p_to_4k"])
def test_true_ratio_presets_whole_frame_per_stage(fsr, parity, name):
    """The per-stage whole-frame classes of test_gpu_fullframe.py::test_whole_frame_two_pass_and_fused on the reference's
    true-ratio presets (FSRSample.h:79-95): EXACT = 0 differing values, F <= 1 ULP and >= 99.5 % bit-equal, fused == two
    dispatches.  Non-2x ratios exercise tap phases, footprint widths
    (54 / 42 texels per 64 columns) and LDS pitches the BASELINE shapes never touch."""
    from test_gpu_fullframe import assert_exact16, assert_f_class, host
    iw, ih, ow, oh = parity.SHAPES[name]
    o = parity.checker()
    img = frames.synthetic_frame(iw, ih, k=1, dtype=np.float16)
    con = o.FsrEasuCon(iw, ih, iw, ih, ow, oh)
    assert np.array_equal(con, fsr.FsrEasuCon(iw, ih, iw, ih, ow, oh))
    rc = o.FsrRcasCon(0.25)
    want_mid = o.easu_f(img.astype(np.float32), ow, oh, con)
    src = torch.from_numpy(img).cuda()
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
        want_out = o.rcas_f(got_mid.astype(np.float32), rc)  # identical input: the GPU's own intermediary
        (assert_exact16 if exact else assert_f_class)(host(out), want_out, tag + " rcas")
        assert torch.equal(out.view(torch.int16), fus.view(torch.int16)), tag + ": fused launch differs from the two dispatches"
    # packed fp16 on the same frame: bit-exact against the CPU-evaluated FsrEasuH
    fsr.easu(src, mid, con=con, flags=fsr.FLAG_MATH_PACKED_FP16)
    assert_exact16(host(mid), o.easu_h(img.astype(np.float32), ow, oh, con), name + " easu H")
